// hip_renderer.cpp — see hip_renderer.h. Plain C++17, no HIP here: everything device-side goes
// through the C ABI of include/ubdpt.h.
#include "hip_renderer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace uhip {

// ---- HipRenderer ------------------------------------------------------------------------------------
HipRenderer::HipRenderer(int device, const ubdpt_options* options) : m_device(device) {
    if (options) m_options = *options;
}

HipRenderer::~HipRenderer() {
    if (m_ctx) ubdpt_destroy(m_ctx);
}

bool HipRenderer::initialize(const RenderParameters& render_params, std::shared_ptr<FlatSceneSource> scene) {
    if (m_ctx) { ubdpt_destroy(m_ctx); m_ctx = nullptr; }
    if (!scene) { m_status = UBDPT_INVALID_SCENE; return false; }
    ubdpt_scene_desc desc;
    std::memset(&desc, 0, sizeof(desc));
    if (!scene->flatten(desc)) { m_status = UBDPT_INVALID_SCENE; return false; }
    // per-pass device buffers for renderPass: two windows of passes rendered ahead (pass_windows.h)
    ubdpt_options opt = m_options;
    if (opt.pass_slots == 0) opt.pass_slots = PassWindows::slotsFor(render_params.img_res_x, render_params.img_res_y);
    opt.pass_slots = std::min<uint32_t>(opt.pass_slots, 64);
    m_status = ubdpt_create(&desc, &render_params, &opt, m_device, &m_ctx);
    m_windows.reset(opt.pass_slots);
    return m_status == UBDPT_SUCCESS;
}

bool HipRenderer::renderPass(std::shared_ptr<SumBuffer> pixel_buffer, size_t curr_pass, size_t num_threads,
                             std::function<void(double)>& update_progress_callback) {
    return renderPassOnto(pixel_buffer.get(), pixel_buffer, curr_pass, num_threads, update_progress_callback);
}

bool HipRenderer::renderPassOnto(const SumBuffer* prev, std::shared_ptr<SumBuffer> pixel_buffer, size_t curr_pass,
                                 size_t num_threads, std::function<void(double)>& update_progress_callback) {
    if (!m_ctx || !pixel_buffer || !prev) { m_status = UBDPT_UNINITIALIZED; return false; }
    m_stop_requested = false;  // UBDPTRenderer::renderPass resets m_stop on entry (ubdptrenderer.cpp:37)
    // this pass is added to the caller's running sum
    m_status = m_windows.servePass(m_ctx, curr_pass, m_seed, prev->ptr(), pixel_buffer->ptr(), (uint32_t)std::min<size_t>(num_threads, 64));
    if (m_status != UBDPT_SUCCESS) return false;
    if (m_stop_requested) {  // stop() arrived during this call
        m_status = UBDPT_STOPPED;
        m_windows.drop();
        return false;
    }
    if (update_progress_callback != nullptr) {
        std::lock_guard<std::mutex> lock(m_progress_mutex);
        update_progress_callback(1.0);
    }
    return true;
}

void HipRenderer::stop() {
    m_stop_requested = true;
    if (m_ctx) ubdpt_request_stop(m_ctx);
}

bool HipRenderer::renderPassesOnDevice(size_t first_pass, size_t n, size_t* done_out,
                                       const std::function<void(double)>* progress) {
    if (done_out) *done_out = 0;
    if (!m_ctx) { m_status = UBDPT_UNINITIALIZED; return false; }
    size_t done_total = 0;
    m_stop_requested = false;  // as renderPass: a stop request refers to the call it arrives in
    if (!progress || !*progress) {
        while (done_total < n) {
            uint32_t chunk = (uint32_t)std::min<size_t>(n - done_total, 1u << 20), done = 0;
            m_status = ubdpt_render_passes(m_ctx, first_pass + done_total, chunk, 1, m_seed, &done);
            done_total += done;
            if (m_status != UBDPT_SUCCESS) break;
            if (m_stop_requested) { m_status = UBDPT_STOPPED; break; }
        }
    } else {
        // With a progress callback the passes are enqueued a few batches at a time (about 16 M pixel-samples per
        // call) and the callback hears, after every call, which fraction of the n passes the GPU has finished
        // (the reference reports finished pixels of its one pass, ubdptrenderer.cpp:54-62).
        uint64_t w = 0, h = 0;
        ubdpt_get_image_size(m_ctx, &w, &h);
        const uint64_t npix = std::max<uint64_t>(1, (uint64_t)w * h);
        const size_t per_call = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(1u << 16, (16ull << 20) / npix));
        uint64_t enq0 = 0, fin = 0;
        ubdpt_query_progress(m_ctx, &enq0, nullptr);
        const double total = (double)n * (double)npix;
        while (done_total < n) {
            uint32_t done = 0;
            m_status = ubdpt_enqueue_passes(m_ctx, first_pass + done_total, (uint32_t)std::min(per_call, n - done_total), 1, m_seed, &done);
            done_total += done;
            if (m_status != UBDPT_SUCCESS) break;
            // every enqueue call clears the context's own stop flag on entry (as UBDPTRenderer::renderPass does), so a
            // request that arrives between two calls, or while a call waits for room on the device, is only seen here
            if (m_stop_requested) { m_status = UBDPT_STOPPED; break; }
            if (ubdpt_query_progress(m_ctx, nullptr, &fin) == UBDPT_SUCCESS && fin >= enq0) {
                std::lock_guard<std::mutex> lock(m_progress_mutex);
                (*progress)(std::min(1.0, (double)(fin - enq0) / total));
            }
        }
        // what was enqueued is finished in any case: the passes counted in done_total are in the device sum
        const int sync = ubdpt_synchronize(m_ctx);
        if (m_status == UBDPT_SUCCESS || (m_status == UBDPT_STOPPED && sync != UBDPT_SUCCESS)) m_status = sync;
        if (m_status == UBDPT_SUCCESS) {
            std::lock_guard<std::mutex> lock(m_progress_mutex);
            (*progress)(1.0);
        }
    }
    if (done_out) *done_out = done_total;
    return m_status == UBDPT_SUCCESS;
}

bool HipRenderer::uploadSum(const SumBuffer* sum) {
    if (!m_ctx) { m_status = UBDPT_UNINITIALIZED; return false; }
    m_status = ubdpt_upload_sum(m_ctx, sum ? sum->ptr() : nullptr);
    return m_status == UBDPT_SUCCESS;
}

bool HipRenderer::readSum(SumBuffer& out, bool add_into) {
    if (!m_ctx) { m_status = UBDPT_UNINITIALIZED; return false; }
    m_status = ubdpt_read_sum(m_ctx, out.ptr(), add_into ? 1 : 0);
    return m_status == UBDPT_SUCCESS;
}

// ---- .ur files -----------------------------------------------------------------------------------------
Result writeUr(const std::string& filename, uint64_t curr_pass, const RenderParameters& params, RendererType type,
               const double* sums) {
    std::ofstream out(filename, std::ios::binary);
    if (!out.is_open()) return Result::UError;
    int32_t t = (int32_t)type;
    out.write(reinterpret_cast<const char*>(&curr_pass), sizeof(curr_pass));
    out.write(reinterpret_cast<const char*>(&params), sizeof(params));
    out.write(reinterpret_cast<const char*>(&t), sizeof(t));
    out.write(reinterpret_cast<const char*>(sums), sizeof(double) * 3 * params.img_res_x * params.img_res_y);
    return out.good() ? Result::USuccess : Result::UError;
}

Result readUr(const std::string& filename, uint64_t& curr_pass, RenderParameters& params, RendererType& type,
              std::vector<double>& sums) {
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) return Result::UError;
    int32_t t = 0;
    in.read(reinterpret_cast<char*>(&curr_pass), sizeof(curr_pass));
    in.read(reinterpret_cast<char*>(&params), sizeof(params));
    in.read(reinterpret_cast<char*>(&t), sizeof(t));
    if (!in.good()) return Result::UInvalidFormat;
    // the reference trusts the header; a size that cannot be allocated or read is reported instead
    if (params.img_res_x == 0 || params.img_res_y == 0 || params.img_res_x > 65536 || params.img_res_y > 65536)
        return Result::UInvalidFormat;
    type = (RendererType)t;
    sums.assign((size_t)3 * params.img_res_x * params.img_res_y, 0.0);
    in.read(reinterpret_cast<char*>(sums.data()), sizeof(double) * sums.size());
    if ((size_t)in.gcount() != sizeof(double) * sums.size()) return Result::UInvalidFormat;
    return Result::USuccess;
}

// ---- HipEngine -------------------------------------------------------------------------------------------
HipEngine::HipEngine(int device, const ubdpt_options* options) : m_device(device) {
    if (options) m_options = *options;
}

bool HipEngine::initPixelBuffers(size_t w, size_t h) {
    m_buffers[0] = std::make_shared<SumBuffer>(w, h);
    m_buffers[1] = std::make_shared<SumBuffer>(w, h);
    return m_buffers[0] && m_buffers[1];
}

Result HipEngine::newRendering(const RenderParameters& params, RendererType type) {
    if (!m_scene) return Result::UInvalidScene;
    m_curr_pass = 0;
    m_params = params;
    m_write = 0; m_read = 1;
    if (!initPixelBuffers(params.img_res_x, params.img_res_y)) return Result::UError;
    // both enumerators are rendered by the HIP renderer here; the value is kept for the .ur header
    if (type != RendererType::BDPT && type != RendererType::BDPT_HIP) return Result::UInvalidFormat;
    m_renderer = std::make_unique<HipRenderer>(m_device, &m_options);
    m_type = type;
    if (!m_renderer->initialize(params, m_scene)) return Result::UError;
    return Result::USuccess;
}

Result HipEngine::saveRendering(const std::string& filename) const {
    if (!m_buffers[m_read]) return Result::UError;
    // The file records WHAT was rendered (the BDPT estimator, sums of passes), not which implementation rendered it:
    // the type written is always BDPT (0), the only value the unmodified reference accepts (uengine.cpp:123-130; it has
    // overwritten its state by the time it rejects an unknown type), so a checkpoint moves between the two freely.
    return writeUr(filename, m_curr_pass, m_params, RendererType::BDPT, m_buffers[m_read]->ptr());
}

Result HipEngine::loadRendering(const std::string& filename, RenderParameters& params, RendererType& type, size_t& curr_pass) {
    if (!m_scene) return Result::UInvalidScene;
    uint64_t pass = 0;
    RenderParameters p{};
    RendererType t{};
    std::vector<double> sums;
    Result r = readUr(filename, pass, p, t, sums);
    if (r != Result::USuccess) return r;
    if (t != RendererType::BDPT && t != RendererType::BDPT_HIP) return Result::UInvalidFormat;
    m_write = 0; m_read = 1;
    m_curr_pass = pass; m_params = p; m_type = t;
    if (!initPixelBuffers(p.img_res_x, p.img_res_y)) return Result::UError;
    std::memcpy(m_buffers[m_read]->ptr(), sums.data(), sizeof(double) * sums.size());
    m_renderer = std::make_unique<HipRenderer>(m_device, &m_options);
    if (!m_renderer->initialize(m_params, m_scene)) return Result::UError;
    params = m_params; curr_pass = m_curr_pass; type = m_type;
    return Result::USuccess;
}

Result HipEngine::renderPass(size_t num_threads, std::function<void(double)> update_progress_callback) {
    if (!m_renderer) return Result::UUninitialized;
    if (num_threads <= 0) num_threads = 1;
    else if (num_threads > kMaxThreads) num_threads = kMaxThreads;
    // the reference copies the read buffer into the write buffer and the renderer adds the pass to it (uengine.cpp:153-164);
    // here write = read + pass in one sweep (the first pass is added to the zeros the write buffer holds)
    const SumBuffer* prev = m_curr_pass != 0 ? m_buffers[m_read].get() : m_buffers[m_write].get();
    bool complete = m_renderer->renderPassOnto(prev, m_buffers[m_write], m_curr_pass, num_threads, update_progress_callback);
    if (complete) {
        m_curr_pass++;
        std::swap(m_read, m_write);
        return Result::USuccess;
    }
    return m_renderer->lastStatus() == UBDPT_STOPPED || m_renderer->lastStatus() == UBDPT_SUCCESS ? Result::UStopped
                                                                                                     : Result::UError;
}

Result HipEngine::renderPasses(size_t n, std::function<void(double)> update_progress_callback) {
    if (!m_renderer) return Result::UUninitialized;
    if (n == 0) return Result::USuccess;
    // device sum := everything rendered so far, then n more passes on top, one read-back at the end
    if (!m_renderer->uploadSum(m_curr_pass != 0 ? m_buffers[m_read].get() : nullptr)) return Result::UError;
    size_t done = 0;
    bool ok = m_renderer->renderPassesOnDevice(m_curr_pass, n, &done, &update_progress_callback);
    if (!ok && m_renderer->lastStatus() != UBDPT_STOPPED) return Result::UError;
    if (done > 0) {
        if (!m_renderer->readSum(*m_buffers[m_write], false)) return Result::UError;
        m_curr_pass += done;
        std::swap(m_read, m_write);
    }
    return ok ? Result::USuccess : Result::UStopped;
}

void HipEngine::stop() {
    if (m_renderer) m_renderer->stop();
}

Result HipEngine::imageRGB(std::vector<double>& img_data, double gamma, size_t& img_width, size_t& img_height) {
    if ((m_curr_pass == 0) || (gamma <= 0) || !m_buffers[m_read] || !m_renderer) return Result::UNoData;
    img_width = m_params.img_res_x;
    img_height = m_params.img_res_y;
    img_data.resize(img_width * img_height * 3);
    // the host buffer is the engine's truth (pass-by-pass rendering leaves only the last pass on the device)
    if (!m_renderer->uploadSum(m_buffers[m_read].get())) return Result::UError;
    int st = ubdpt_image_rgb(m_renderer->context(), m_curr_pass, gamma, img_data.data());
    if (st == UBDPT_NO_DATA) return Result::UNoData;
    return st == UBDPT_SUCCESS ? Result::USuccess : Result::UError;
}

}  // namespace uhip

// ---- C shim over the classes (used by the Python tests; a C++ caller uses the classes directly) ---------
extern "C" {

struct uhip_engine {
    uhip::HipEngine engine;
    uhip_engine(int device, const ubdpt_options* o) : engine(device, o) {}
};

uhip_engine* uhip_engine_create(int device, const ubdpt_options* options) { return new uhip_engine(device, options); }
void uhip_engine_destroy(uhip_engine* e) { delete e; }
void uhip_engine_set_scene(uhip_engine* e, const ubdpt_scene_desc* desc) {
    e->engine.setScene(std::make_shared<uhip::BorrowedScene>(*desc));
}
int uhip_engine_new_rendering(uhip_engine* e, const ubdpt_render_params* p, int type) {
    return (int)e->engine.newRendering(*p, (uhip::RendererType)type);
}
int uhip_engine_render_pass(uhip_engine* e, size_t num_threads, void (*progress)(double)) {
    std::function<void(double)> cb;
    if (progress) cb = progress;
    return (int)e->engine.renderPass(num_threads, cb);
}
int uhip_engine_render_passes(uhip_engine* e, size_t n) { return (int)e->engine.renderPasses(n); }
int uhip_engine_render_passes_progress(uhip_engine* e, size_t n, void (*progress)(double)) {
    std::function<void(double)> cb;
    if (progress) cb = progress;
    return (int)e->engine.renderPasses(n, cb);
}
void uhip_engine_stop(uhip_engine* e) { e->engine.stop(); }
size_t uhip_engine_curr_pass(uhip_engine* e) { return e->engine.currPass(); }
int uhip_engine_sum(uhip_engine* e, double* out) {
    const uhip::SumBuffer* s = e->engine.sum();
    if (!s || !out) return (int)uhip::Result::UNoData;
    std::memcpy(out, s->ptr(), sizeof(double) * 3 * s->width() * s->height());
    return (int)uhip::Result::USuccess;
}
int uhip_engine_image_rgb(uhip_engine* e, double gamma, double* out, size_t capacity) {
    std::vector<double> img;
    size_t w = 0, h = 0;
    int r = (int)e->engine.imageRGB(img, gamma, w, h);
    if (r != 0) return r;
    if (!out || capacity < img.size()) return (int)uhip::Result::UError;
    std::memcpy(out, img.data(), sizeof(double) * img.size());
    return r;
}
int uhip_engine_save(uhip_engine* e, const char* filename) { return (int)e->engine.saveRendering(filename); }
int uhip_engine_load(uhip_engine* e, const char* filename, ubdpt_render_params* params, int* type, size_t* curr_pass) {
    uhip::RendererType t{};
    int r = (int)e->engine.loadRendering(filename, *params, t, *curr_pass);
    *type = (int)t;
    return r;
}
int uhip_ur_write(const char* filename, uint64_t curr_pass, const ubdpt_render_params* params, int type, const double* sums) {
    return (int)uhip::writeUr(filename, curr_pass, *params, (uhip::RendererType)type, sums);
}
// sums may be NULL to read the header only; otherwise it must hold 3 * W * H doubles
int uhip_ur_read(const char* filename, uint64_t* curr_pass, ubdpt_render_params* params, int* type, double* sums,
                 size_t sums_capacity) {
    std::vector<double> v;
    uhip::RendererType t{};
    int r = (int)uhip::readUr(filename, *curr_pass, *params, t, v);
    if (r != 0) return r;
    *type = (int)t;
    if (sums) {
        if (sums_capacity < v.size()) return (int)uhip::Result::UError;
        std::memcpy(sums, v.data(), sizeof(double) * v.size());
    }
    return r;
}

// Times the loop the reference's engine runs around the URenderer subclass of host/reference_adapter/
// (uengine/uengine.cpp:144-180: copy the read buffer into the write buffer, renderPass adds one pass to it, swap), with
// the calls that subclass makes: PassWindows::slotsFor at create, PassWindows::servePass(sum, sum) per pass. `warmup`
// passes first, then `passes` timed ones; engine_copy = 0 leaves out the engine's own copy (what the renderer alone
// costs). Returns the seconds of the timed passes, negative on failure.
double uhip_adapter_pass_rate(const ubdpt_scene_desc* desc, const ubdpt_render_params* params, int device, size_t warmup,
                              size_t passes, size_t num_threads, int engine_copy) {
    if (!desc || !params) return -1.0;
    ubdpt_options opt{};
    opt.pass_slots = uhip::PassWindows::slotsFor(params->img_res_x, params->img_res_y);
    ubdpt_context* ctx = nullptr;
    if (ubdpt_create(desc, params, &opt, device, &ctx) != UBDPT_SUCCESS) return -1.0;
    uhip::PassWindows windows;
    windows.reset(opt.pass_slots);
    uhip::SumBuffer a(params->img_res_x, params->img_res_y), b(params->img_res_x, params->img_res_y);
    uhip::SumBuffer* rd = &a;
    uhip::SumBuffer* wr = &b;
    double seconds = -1.0;
    std::chrono::steady_clock::time_point t0;
    for (size_t pass = 0; pass < warmup + passes; pass++) {
        if (pass == warmup) t0 = std::chrono::steady_clock::now();
        if (engine_copy && pass != 0) wr->setFrom(*rd);                                  // uengine.cpp:153-156
        if (windows.servePass(ctx, pass, 0x5EED, wr->ptr(), wr->ptr(), (uint32_t)num_threads) != UBDPT_SUCCESS) { seconds = -2.0; break; }
        std::swap(rd, wr);                                                               // :167-174
        if (pass + 1 == warmup + passes) seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    ubdpt_destroy(ctx);
    return seconds;
}

}  // extern "C"
