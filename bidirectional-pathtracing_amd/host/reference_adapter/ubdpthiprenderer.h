// ubdpthiprenderer.h — file a reference maintainer adds to uengine/: the URenderer (uengine/urenderer.h:10-16)
// that forwards to the MI355X backend through the C ABI (include/ubdpt.h). Compiles against the reference's own
// headers; the build of this repository only syntax-checks it (tests/test_adapter_compiles.py), it is not part of
// any library here. uhip::HipRenderer (../hip_renderer.cpp) is the same logic against GLM-free mirror types.
#ifndef UBDPTHIPRENDERER_H
#define UBDPTHIPRENDERER_H

#include "urenderer.h"
#include "uscene.h"
#include "uflatsceneexport.h"

#include <ubdpt.h>
#include <mutex>

class UBDPTHipRenderer : public URenderer
{
public:
    ~UBDPTHipRenderer() { if(m_ctx) ubdpt_destroy(m_ctx); }

    bool initialize(const URenderParameters& params, std::shared_ptr<UScene> scene) override
    {
        auto* exporter = dynamic_cast<UFlatSceneExport*>(scene.get());
        if(exporter == nullptr) return false;
        ubdpt_scene_desc desc{};
        if(!exporter->flatten(desc)) return false;
        static_assert(sizeof(URenderParameters) == sizeof(ubdpt_render_params), "uutils.h:11-20");
        if(m_ctx) { ubdpt_destroy(m_ctx); m_ctx = nullptr; }
        ubdpt_options opt{}; opt.pass_slots = 2 * kSlots; m_win_n = m_next_n = 0;
        return ubdpt_create(&desc, reinterpret_cast<const ubdpt_render_params*>(&params), &opt, /*device*/0, &m_ctx) == UBDPT_SUCCESS;
    }

    // The buffer already holds the sum of all previous passes (uengine.cpp:153-156); add exactly one
    // sample per pixel for pass curr_pass. The device renders; num_threads CPU threads add the pass to the buffer.
    bool renderPass(std::shared_ptr<UBuffer2D<glm::dvec3>> pixel_buffer, size_t curr_pass, size_t num_threads,
                    std::function<void(double)>& update_progress_callback) override
    {
        // One pass is a small batch for the device, so a window of kSlots consecutive passes is rendered at once
        // into per-pass device buffers and served call by call; the device holds two windows (pass_slots =
        // 2 * kSlots at create) and renders the next one while the host consumes the current one.
        if(!(m_win_n && curr_pass >= m_win_first && curr_pass < m_win_first + m_win_n))
        {
            uint32_t done = 0;
            if(m_next_n && curr_pass >= m_next_first && curr_pass < m_next_first + m_next_n)
            {   m_win_first = m_next_first; m_win_n = m_next_n; m_half ^= 1u; }           // already enqueued
            else
            {
                m_win_n = 0; m_half = 0;
                if(ubdpt_enqueue_passes_separate(m_ctx, curr_pass, kSlots, 1, m_seed, 0, &done) != UBDPT_SUCCESS || done == 0)
                    return false;                                                      // stopped: UEngine drops the buffer (:176-179)
                m_win_first = curr_pass; m_win_n = done;
            }
            m_next_n = 0;
            if(m_win_n == kSlots && ubdpt_enqueue_passes_separate(m_ctx, m_win_first + kSlots, kSlots, 1, m_seed, (m_half ^ 1u) * kSlots, &done) == UBDPT_SUCCESS)
            {   m_next_first = m_win_first + kSlots; m_next_n = done; }
        }
        // UBuffer2D<glm::dvec3> is contiguous [y][x] of 3 doubles (uutils.h:37-98)
        double* sum = reinterpret_cast<double*>(pixel_buffer->ptr());  // += of the pass on the threads this call was given
        if(ubdpt_read_sum_slot_onto(m_ctx, m_half * kSlots + uint32_t(curr_pass - m_win_first), sum, sum, uint32_t(num_threads)) != UBDPT_SUCCESS) return false;
        if(update_progress_callback != nullptr) { std::lock_guard<std::mutex> l(m_progress); update_progress_callback(1.0); }
        return true;
    }

    void stop() override { if(m_ctx) ubdpt_request_stop(m_ctx); }                      // any thread (appmanager.cpp:489-498)

private:
    ubdpt_context* m_ctx = nullptr;
    static constexpr uint32_t kSlots = 16;        // about 4 M pixel-samples per window at 512 x 512
    size_t m_win_first = 0, m_win_n = 0, m_next_first = 0, m_next_n = 0;   // reset in initialize()
    uint32_t m_half = 0;                           // which half of the per-pass buffers the current window is in
    uint64_t m_seed = 0x5EED;   // the reference seeds from std::random_device (umath.cpp:10-20); any value works
    std::mutex m_progress;
};

#endif // UBDPTHIPRENDERER_H
