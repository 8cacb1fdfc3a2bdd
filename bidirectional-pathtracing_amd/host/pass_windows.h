// pass_windows.h — serves URenderer::renderPass (uengine/urenderer.h:13-14; one pass per call, added to the caller's
// host buffer, uengine/uengine.cpp:153-164) from windows of passes the device renders ahead.
//
// One pass is a small batch for the device (a fraction of a millisecond of work behind two dozen launches), so a window
// of consecutive passes is rendered at once into per-pass device sum buffers (ubdpt_enqueue_passes_separate) and the
// following calls are served from it (ubdpt_read_sum_slot_onto); the context holds two windows of buffers, and as soon
// as one becomes current the passes right after it are enqueued into the other half, so the device renders the next
// window while the host consumes this one. A call for a pass outside both windows (new rendering, resume elsewhere,
// other seed) starts over; a stop or a jump wastes at most one window of device work. The samples of a pass do not
// depend on what it is batched with, so the engine gets exactly the passes it asked for.
//
// Depends on the C ABI only (no GLM, no engine types): uhip::HipRenderer (hip_renderer.cpp) and the URenderer subclass a
// maintainer adds to the reference (reference_adapter/ubdpthiprenderer.h) both delegate to this one class.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "../../include/ubdpt.h"

namespace uhip {

class PassWindows {
public:
    // Per-pass buffers to ask ubdpt_create for (ubdpt_options.pass_slots): two windows of about 16 M pixel-samples, at
    // most 32 passes each (measured pass by pass: 2 / 4 / 8 passes per window at 1080p on a 2 M-triangle scene
    // 243 / 259 / 282 Mpaths/s, 16 / 32 passes at 512 x 512 359 / 402).
    static uint32_t slotsFor(uint64_t img_res_x, uint64_t img_res_y) {
        const uint64_t npix = img_res_x * img_res_y;
        return 2u * (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(32, npix ? (1ull << 24) / npix : 1));
    }
    // After ubdpt_create: `pass_slots` = what the context was created with (1: a single window of one pass).
    void reset(uint32_t pass_slots) {
        pass_slots = std::min<uint32_t>(std::max<uint32_t>(pass_slots, 1), 64);
        m_two = pass_slots >= 2;
        m_slots = m_two ? pass_slots / 2 : 1;
        drop();
    }
    void drop() { m_cur = Window{}; m_next = Window{}; }
    uint32_t passesPerWindow() const { return m_slots; }

    // out = prev + pass `curr_pass` of the rendering with `seed` (prev may be out), summed on `host_threads` CPU threads.
    // UBDPT_SUCCESS, UBDPT_STOPPED when a stop request kept the window from being enqueued, or the failing status.
    int servePass(ubdpt_context* ctx, size_t curr_pass, uint64_t seed, const double* prev, double* out, uint32_t host_threads) {
        if (!m_cur.holds(curr_pass, seed)) {
            if (m_next.holds(curr_pass, seed)) {
                m_cur = m_next;
                m_next = Window{};
            } else {
                m_next = Window{};
                const int st = enqueue(ctx, m_cur, curr_pass, seed, 0);
                if (st != UBDPT_SUCCESS) return st;
            }
            // the passes after this window go to the other half of the buffers right away
            if (m_two && m_cur.count == m_slots) (void)enqueue(ctx, m_next, m_cur.first + m_cur.count, seed, m_cur.half ^ 1u);
        }
        // waits for the window the pass belongs to, not for the next one
        return ubdpt_read_sum_slot_onto(ctx, m_cur.half * m_slots + (uint32_t)(curr_pass - m_cur.first), prev, out,
                                        std::min<uint32_t>(std::max<uint32_t>(host_threads, 1), 16));
    }

private:
    struct Window {
        size_t first = 0, count = 0;
        uint64_t seed = 0;
        uint32_t half = 0;
        bool holds(size_t pass, uint64_t s) const { return count != 0 && seed == s && pass >= first && pass < first + count; }
    };
    int enqueue(ubdpt_context* ctx, Window& w, size_t first_pass, uint64_t seed, uint32_t half) {
        w = Window{};
        uint32_t done = 0;
        const int st = ubdpt_enqueue_passes_separate(ctx, first_pass, m_slots, 1, seed, half * m_slots, &done);
        if (st != UBDPT_SUCCESS) return st;
        if (done == 0) return UBDPT_STOPPED;   // stopped before the first pass: the engine drops its write buffer
        w.first = first_pass; w.count = done; w.seed = seed; w.half = half;
        return UBDPT_SUCCESS;
    }
    uint32_t m_slots = 1;   // passes per window
    bool m_two = false;     // the context holds 2 * m_slots per-pass buffers
    Window m_cur, m_next;
};

}  // namespace uhip
