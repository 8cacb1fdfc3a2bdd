// ubdpthiprenderer.h — file a reference maintainer adds to uengine/: the URenderer (uengine/urenderer.h:10-16)
// that forwards to the MI355X backend through the C ABI (include/ubdpt.h). Compiles against the reference's own
// headers; the build of this repository only syntax-checks it (tests/test_adapter_compiles.py), it is not part of
// any library here. The serving of passes from windows rendered ahead is uhip::PassWindows (../pass_windows.h), the
// same object uhip::HipRenderer (../hip_renderer.cpp, the GLM-free mirror this repository tests and times) delegates to.
#ifndef UBDPTHIPRENDERER_H
#define UBDPTHIPRENDERER_H

#include "urenderer.h"
#include "uscene.h"
#include "uflatsceneexport.h"

#include <ubdpt.h>
#include <pass_windows.h>   // bidirectional-pathtracing_amd/host/pass_windows.h: needs <ubdpt.h> only
#include <atomic>
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
        // per-pass device buffers: two windows of passes rendered ahead, sized by the resolution (pass_windows.h)
        ubdpt_options opt{};
        opt.pass_slots = uhip::PassWindows::slotsFor(params.img_res_x, params.img_res_y);
        const int st = ubdpt_create(&desc, reinterpret_cast<const ubdpt_render_params*>(&params), &opt, /*device*/0, &m_ctx);
        m_windows.reset(opt.pass_slots);
        return st == UBDPT_SUCCESS;
    }

    // The buffer already holds the sum of all previous passes (uengine.cpp:153-156); add exactly one
    // sample per pixel for pass curr_pass. The device renders; num_threads CPU threads add the pass to the buffer.
    bool renderPass(std::shared_ptr<UBuffer2D<glm::dvec3>> pixel_buffer, size_t curr_pass, size_t num_threads,
                    std::function<void(double)>& update_progress_callback) override
    {
        if(m_ctx == nullptr || pixel_buffer == nullptr) return false;
        m_stop = false;                                                                 // as ubdptrenderer.cpp:37
        // UBuffer2D<glm::dvec3> is contiguous [y][x] of 3 doubles (uutils.h:37-98): sum += the pass
        double* sum = reinterpret_cast<double*>(pixel_buffer->ptr());
        if(m_windows.servePass(m_ctx, curr_pass, m_seed, sum, sum, uint32_t(num_threads)) != UBDPT_SUCCESS)
            return false;                                                               // stopped: UEngine drops the buffer (:176-179)
        if(m_stop) { m_windows.drop(); return false; }                                  // stop() arrived during this call
        if(update_progress_callback != nullptr) { std::lock_guard<std::mutex> l(m_progress); update_progress_callback(1.0); }
        return true;
    }

    void stop() override { m_stop = true; if(m_ctx) ubdpt_request_stop(m_ctx); }        // any thread (appmanager.cpp:489-498)

private:
    ubdpt_context* m_ctx = nullptr;
    uhip::PassWindows m_windows;
    std::atomic<bool> m_stop{false};
    uint64_t m_seed = 0x5EED;   // the reference seeds from std::random_device (umath.cpp:10-20); any value works
    std::mutex m_progress;
};

#endif // UBDPTHIPRENDERER_H
