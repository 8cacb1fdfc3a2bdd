// multi_gpu.cpp — see multi_gpu.h. Host code only: HIP runtime calls for buffers, RCCL for the reduce.
#include "multi_gpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace uhip {

namespace {
bool hipOk(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "uhip: %s: %s\n", what, hipGetErrorString(e));
    return false;
}
bool ncclOk(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    fprintf(stderr, "uhip: %s: %s\n", what, ncclGetErrorString(r));
    return false;
}
}  // namespace

int reduceSumRccl(ubdpt_context* ctx, void* nccl_comm, int root, double* device_buffer, void* hip_stream) {
    if (!ctx || !nccl_comm || !device_buffer) return UBDPT_INVALID_ARG;
    int s = ubdpt_read_sum_device(ctx, device_buffer);  // waits for the context's passes, interleaves to [y][x][rgb]
    if (s != UBDPT_SUCCESS) return s;
    uint64_t w = 0, h = 0;
    s = ubdpt_get_image_size(ctx, &w, &h);
    if (s != UBDPT_SUCCESS) return s;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (!ncclOk(ncclReduce(device_buffer, device_buffer, (size_t)(w * h * 3), ncclDouble, ncclSum, root,
                           static_cast<ncclComm_t>(nccl_comm), stream), "ncclReduce"))
        return UBDPT_HIP_ERROR;
    if (!hipOk(hipStreamSynchronize(stream), "hipStreamSynchronize")) return UBDPT_HIP_ERROR;
    return UBDPT_SUCCESS;
}

MultiGpuRenderer::MultiGpuRenderer(std::vector<int> devices, const ubdpt_options* options) : m_devices(std::move(devices)) {
    if (options) m_options = *options;
    for (int d : m_devices)
        if (std::find(m_unique.begin(), m_unique.end(), d) == m_unique.end()) m_unique.push_back(d);
}

MultiGpuRenderer::~MultiGpuRenderer() { release(); }

void MultiGpuRenderer::release() {
    for (ubdpt_context* c : m_ctx) ubdpt_destroy(c);
    m_ctx.clear();
    for (size_t r = 0; r < m_partial.size(); r++) {
        if (hipSetDevice(m_unique[r]) == hipSuccess) (void)hipFree(m_partial[r]);
    }
    m_partial.clear();
    for (void* c : m_comms) ncclCommDestroy(static_cast<ncclComm_t>(c));
    m_comms.clear();
}

bool MultiGpuRenderer::initialize(const RenderParameters& params, const ubdpt_scene_desc& scene) {
    release();
    if (m_devices.empty()) { m_status = UBDPT_INVALID_ARG; return false; }
    m_width = params.img_res_x; m_height = params.img_res_y;
    // contexts that share a device split its memory: each may size its batches from an equal share
    for (size_t k = 0; k < m_devices.size(); k++) {
        ubdpt_context* c = nullptr;
        m_status = ubdpt_create(&scene, &params, &m_options, m_devices[k], &c);
        if (m_status != UBDPT_SUCCESS) { release(); return false; }
        m_ctx.push_back(c);
    }
    const size_t bytes = sizeof(double) * 3 * m_width * m_height;
    for (int d : m_unique) {
        double* p = nullptr;
        if (!hipOk(hipSetDevice(d), "hipSetDevice") || !hipOk(hipMalloc((void**)&p, bytes), "hipMalloc")) {
            m_status = UBDPT_HIP_ERROR; release(); return false;
        }
        m_partial.push_back(p);
    }
    std::vector<ncclComm_t> comms(m_unique.size());
    if (!ncclOk(ncclCommInitAll(comms.data(), (int)m_unique.size(), m_unique.data()), "ncclCommInitAll")) {
        m_status = UBDPT_HIP_ERROR; release(); return false;
    }
    for (ncclComm_t c : comms) m_comms.push_back(c);
    return true;
}

bool MultiGpuRenderer::renderPasses(size_t first_pass, size_t n, uint64_t seed, size_t* done) {
    if (done) *done = 0;
    if (m_ctx.empty()) { m_status = UBDPT_UNINITIALIZED; return false; }
    const size_t K = m_ctx.size();
    size_t total = 0;
    bool ok = true;
    // enqueue on every context (the calls return once the batches are queued), then wait for all of them
    for (size_t k = 0; k < K; k++) {
        const size_t count = n > k ? (n - k + K - 1) / K : 0;
        if (count == 0) continue;
        uint32_t q = 0;
        int s = ubdpt_enqueue_passes(m_ctx[k], first_pass + k, (uint32_t)count, K, seed, &q);
        total += q;
        if (s != UBDPT_SUCCESS) { m_status = s; ok = false; }
    }
    for (ubdpt_context* c : m_ctx) {
        int s = ubdpt_synchronize(c);
        if (s != UBDPT_SUCCESS) { m_status = s; ok = false; }
    }
    if (done) *done = total;
    return ok;
}

const double* MultiGpuRenderer::reduceSumOnDevice() {
    if (m_ctx.empty()) { m_status = UBDPT_UNINITIALIZED; return nullptr; }
    const size_t count = 3 * m_width * m_height;
    // per device: the first context's sum replaces the partial buffer, the others are added to it
    for (size_t r = 0; r < m_unique.size(); r++) {
        bool first = true;
        for (size_t k = 0; k < m_ctx.size(); k++) {
            if (m_devices[k] != m_unique[r]) continue;
            int s = first ? ubdpt_read_sum_device(m_ctx[k], m_partial[r]) : ubdpt_accumulate_sum_device(m_ctx[k], m_partial[r]);
            first = false;
            if (s != UBDPT_SUCCESS) { m_status = s; return nullptr; }
        }
    }
    // one reduce(sum) to the first device; a single process drives all ranks, so the calls form one group
    if (!ncclOk(ncclGroupStart(), "ncclGroupStart")) { m_status = UBDPT_HIP_ERROR; return nullptr; }
    for (size_t r = 0; r < m_unique.size(); r++) {
        if (!hipOk(hipSetDevice(m_unique[r]), "hipSetDevice") ||
            !ncclOk(ncclReduce(m_partial[r], m_partial[r], count, ncclDouble, ncclSum, 0, static_cast<ncclComm_t>(m_comms[r]), nullptr),
                    "ncclReduce")) {
            ncclGroupEnd();
            m_status = UBDPT_HIP_ERROR;
            return nullptr;
        }
    }
    if (!ncclOk(ncclGroupEnd(), "ncclGroupEnd")) { m_status = UBDPT_HIP_ERROR; return nullptr; }
    for (size_t r = 0; r < m_unique.size(); r++)
        if (!hipOk(hipSetDevice(m_unique[r]), "hipSetDevice") || !hipOk(hipStreamSynchronize(nullptr), "hipStreamSynchronize")) {
            m_status = UBDPT_HIP_ERROR; return nullptr;
        }
    if (!hipOk(hipSetDevice(m_unique[0]), "hipSetDevice")) { m_status = UBDPT_HIP_ERROR; return nullptr; }
    return m_partial[0];
}

bool MultiGpuRenderer::reduceSumTo(double* host_rgb) {
    if (!host_rgb) { m_status = UBDPT_INVALID_ARG; return false; }
    const double* total = reduceSumOnDevice();
    if (!total) return false;
    // the only copy of the image: device 0 -> the caller's buffer
    if (!hipOk(hipMemcpy(host_rgb, total, sizeof(double) * 3 * m_width * m_height, hipMemcpyDeviceToHost), "hipMemcpy")) {
        m_status = UBDPT_HIP_ERROR; return false;
    }
    return true;
}

bool MultiGpuRenderer::reduceSum(SumBuffer& out) {
    if (out.width() != m_width || out.height() != m_height) { m_status = UBDPT_INVALID_ARG; return false; }
    return reduceSumTo(out.ptr());
}

void MultiGpuRenderer::clear() {
    for (ubdpt_context* c : m_ctx) ubdpt_upload_sum(c, nullptr);
}

void MultiGpuRenderer::stop() {
    for (ubdpt_context* c : m_ctx) ubdpt_request_stop(c);
}

}  // namespace uhip

extern "C" {

struct uhip_multi {
    uhip::MultiGpuRenderer r;
    size_t w = 0, h = 0;
    uhip_multi(std::vector<int> d, const ubdpt_options* o) : r(std::move(d), o) {}
};

uhip_multi* uhip_multi_create(const ubdpt_scene_desc* scene, const ubdpt_render_params* params, const ubdpt_options* options,
                              const int* devices, size_t n_devices) {
    if (!scene || !params || !devices || n_devices == 0) return nullptr;
    uhip_multi* m = new uhip_multi(std::vector<int>(devices, devices + n_devices), options);
    if (!m->r.initialize(*params, *scene)) { delete m; return nullptr; }
    m->w = params->img_res_x; m->h = params->img_res_y;
    return m;
}
void uhip_multi_destroy(uhip_multi* m) { delete m; }
int uhip_multi_render_passes(uhip_multi* m, size_t first_pass, size_t n, uint64_t seed, size_t* done) {
    if (!m) return UBDPT_INVALID_ARG;
    return m->r.renderPasses(first_pass, n, seed, done) ? UBDPT_SUCCESS : m->r.lastStatus();
}
int uhip_multi_reduce_sum(uhip_multi* m, double* rgb) {
    if (!m || !rgb) return UBDPT_INVALID_ARG;
    return m->r.reduceSumTo(rgb) ? UBDPT_SUCCESS : m->r.lastStatus();
}
void uhip_multi_clear(uhip_multi* m) { if (m) m->r.clear(); }
int uhip_reduce_sum_rccl(ubdpt_context* ctx, void* nccl_comm, int root, double* device_buffer, void* hip_stream) {
    return uhip::reduceSumRccl(ctx, nccl_comm, root, device_buffer, hip_stream);
}

}  // extern "C"
