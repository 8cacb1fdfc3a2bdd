// hip_renderer.h — C++ host layer above the C ABI (include/ubdpt.h): the reference's renderer and
// engine interfaces, re-stated for the HIP backend.
//
//   HipRenderer  mirrors URenderer (uengine/urenderer.h:10-16): initialize / renderPass / stop with
//                the reference's semantics — renderPass receives a buffer that already holds the sum
//                of all previous passes and ADDS one sample per pixel for pass `curr_pass`
//                (uengine/uengine.cpp:153-164, uengine/ubdptrenderer.cpp:35-81), returns false when
//                stopped.
//   HipEngine    mirrors UEngine (uengine/uengine.h:21-63, uengine/uengine.cpp): scene, pass counter,
//                double sum buffer with swap-on-success, .ur save / load, stop; plus renderPasses(n),
//                the batched fast path that keeps the sum on the device.
//
// The reference's own types need GLM, which this repository does not ship; the mirrors below have
// the same memory layout (SumBuffer = UBuffer2D<glm::dvec3>, RenderParameters = URenderParameters),
// so the glue a reference maintainer adds is a pointer cast (see the integration notes).
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ubdpt.h"
#include "pass_windows.h"

namespace uhip {

// UResult (uengine/uengine.h:18), same order
enum class Result { USuccess, UUninitialized, UInvalidScene, UInvalidFormat, UStopped, UNoData, UError };

// URendererType (uengine/uengine.h:19) with the new enumerator APPENDED: the value is written raw
// into .ur files (uengine/uengine.cpp:78), so BDPT must stay 0.
enum class RendererType : int32_t { BDPT = 0, BDPT_HIP = 1 };

using RenderParameters = ubdpt_render_params;  // URenderParameters, uengine/uutils.h:11-20 (56 bytes)

// UBuffer2D<glm::dvec3> (uengine/uutils.h:37-98): contiguous row-major [y][x] of three doubles.
class SumBuffer {
public:
    SumBuffer(size_t width, size_t height) : m_width(width), m_height(height), m_data(width * height * 3, 0.0) {}
    size_t width() const { return m_width; }
    size_t height() const { return m_height; }
    double* ptr() { return m_data.data(); }
    const double* ptr() const { return m_data.data(); }
    double* at(size_t x, size_t y) { return &m_data[(y * m_width + x) * 3]; }
    void setFrom(const SumBuffer& o) { m_data = o.m_data; }  // UBuffer2D::setFrom, uutils.h:75-84
private:
    size_t m_width, m_height;
    std::vector<double> m_data;
};

// The additive scene-export hook: UScene exposes geometry only through opaque virtuals
// (uengine/uscene.h:16-18), so a scene that wants the HIP renderer also implements this.
// The description must stay valid until initialize() returns (it is copied to the device there).
class FlatSceneSource {
public:
    virtual ~FlatSceneSource() = default;
    virtual bool flatten(ubdpt_scene_desc& out) = 0;
};

// A FlatSceneSource over a caller-owned description.
class BorrowedScene : public FlatSceneSource {
public:
    explicit BorrowedScene(const ubdpt_scene_desc& d) : m_desc(d) {}
    bool flatten(ubdpt_scene_desc& out) override { out = m_desc; return true; }
private:
    ubdpt_scene_desc m_desc;
};

class HipRenderer {
public:
    explicit HipRenderer(int device = 0, const ubdpt_options* options = nullptr);
    ~HipRenderer();
    HipRenderer(const HipRenderer&) = delete;
    HipRenderer& operator=(const HipRenderer&) = delete;

    // URenderer::initialize (uengine/ubdptrenderer.cpp:7-33)
    bool initialize(const RenderParameters& render_params, std::shared_ptr<FlatSceneSource> scene);
    // URenderer::renderPass (uengine/ubdptrenderer.cpp:35-81). num_threads CPU threads (at most 16) add the finished
    // pass to the buffer; the rendering itself runs on the device.
    bool renderPass(std::shared_ptr<SumBuffer> pixel_buffer, size_t curr_pass, size_t num_threads,
                    std::function<void(double)>& update_progress_callback);
    // The same with the sum so far in another buffer: out = prev + the pass, in one sweep over the image. This is what
    // UEngine::renderPass does in two (copy of the read buffer into the write buffer, uengine/uengine.cpp:153-156, then
    // the renderer's +=); prev == out.get() is renderPass.
    bool renderPassOnto(const SumBuffer* prev, std::shared_ptr<SumBuffer> out, size_t curr_pass, size_t num_threads,
                        std::function<void(double)>& update_progress_callback);
    // URenderer::stop (uengine/ubdptrenderer.cpp:83-86); callable from another thread
    void stop();

    // Batched form: passes [first_pass, first_pass + n) accumulate in the device sum buffer; nothing
    // crosses PCIe until readSum. `done` receives the passes finished when a stop arrived. With a progress callback
    // the passes are enqueued a few batches at a time and the callback is told the finished fraction of the n passes.
    bool renderPassesOnDevice(size_t first_pass, size_t n, size_t* done = nullptr,
                              const std::function<void(double)>* progress = nullptr);
    bool uploadSum(const SumBuffer* sum);              // nullptr clears the device sum
    bool readSum(SumBuffer& out, bool add_into);
    void setSeed(uint64_t seed) { m_seed = seed; }
    int lastStatus() const { return m_status; }
    ubdpt_context* context() { return m_ctx; }

private:
    int m_device;
    ubdpt_options m_options{};
    ubdpt_context* m_ctx = nullptr;
    uint64_t m_seed = 0x5EED;
    int m_status = UBDPT_SUCCESS;
    std::mutex m_progress_mutex;  // the reference serialises its progress callback (ubdptrenderer.cpp:54-62)
    PassWindows m_windows;        // renderPass is served from windows of passes rendered ahead (pass_windows.h)
    std::atomic<bool> m_stop_requested{false};
};

class HipEngine {
public:
    explicit HipEngine(int device = 0, const ubdpt_options* options = nullptr);

    void setScene(std::shared_ptr<FlatSceneSource> scene) { m_scene = std::move(scene); }   // uengine.cpp:15-20
    Result newRendering(const RenderParameters& params, RendererType type = RendererType::BDPT_HIP);  // :33-63
    Result saveRendering(const std::string& filename) const;                                          // :65-86
    Result loadRendering(const std::string& filename, RenderParameters& params, RendererType& type, size_t& curr_pass);  // :88-142
    Result renderPass(size_t num_threads, std::function<void(double)> update_progress_callback);      // :144-180
    // n passes with the sum resident on the device; the host sum buffer is refreshed once at the end
    Result renderPasses(size_t n, std::function<void(double)> update_progress_callback = nullptr);
    void stop();                                                                                      // :204-208
    // :182-202 with URgbFormat::sRGB: H x W x 3 display values in [0,1]; the conversion runs on the device
    Result imageRGB(std::vector<double>& img_data, double gamma, size_t& img_width, size_t& img_height);
    size_t currPass() const { return m_curr_pass; }
    const SumBuffer* sum() const { return m_buffers[m_read].get(); }   // radiance = sum / currPass (uengine.cpp:197)
    HipRenderer* renderer() { return m_renderer.get(); }

private:
    bool initPixelBuffers(size_t w, size_t h);
    int m_device;
    ubdpt_options m_options{};
    std::shared_ptr<FlatSceneSource> m_scene;
    std::unique_ptr<HipRenderer> m_renderer;
    RenderParameters m_params{};
    RendererType m_type = RendererType::BDPT_HIP;
    size_t m_curr_pass = 0;
    std::shared_ptr<SumBuffer> m_buffers[2];
    size_t m_write = 0, m_read = 1;
    static constexpr size_t kMaxThreads = 64;  // uengine/uengine.h:59
};

// .ur checkpoint (uengine/uengine.cpp:65-142): raw size_t pass counter, raw URenderParameters, raw
// URendererType, then W*H raw dvec3 sums — 68 header bytes, no magic, host endianness.
constexpr size_t kUrHeaderBytes = sizeof(uint64_t) + sizeof(RenderParameters) + sizeof(int32_t);
Result writeUr(const std::string& filename, uint64_t curr_pass, const RenderParameters& params, RendererType type,
               const double* sums);
Result readUr(const std::string& filename, uint64_t& curr_pass, RenderParameters& params, RendererType& type,
              std::vector<double>& sums);

}  // namespace uhip
