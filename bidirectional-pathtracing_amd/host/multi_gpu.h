// multi_gpu.h — sample passes partitioned over several GPUs of one node, combined with an RCCL reduce.
//
// The reference renders a pass with N threads that all add into one pixel buffer
// (uengine/ubdptrenderer.cpp:43-75, :144, :153). Passes are independent samples that meet only in that +=, and the
// pass index only selects the stratum (:95-96), so here context k of K renders passes first + k, first + k + K, ...
// of the whole image into its own device sum buffer, with no communication while rendering; at the end the
// buffers are summed: contexts that share a device are added on that device, the per-device partial sums are
// combined by one ncclReduce(sum) over xGMI (RCCL), and the radiance is the reduced sum / total passes
// (uengine/uengine.cpp:197).
//
// Two front ends:
//   MultiGpuRenderer   one process drives all GPUs (its own communicator, ncclCommInitAll)
//   reduceSumRccl      one process per GPU (the caller brings the communicator it made from its launcher's
//                      rendezvous, e.g. MPI or torch.distributed): the reduce of one context's sum
#pragma once

#include <memory>
#include <vector>

#include "hip_renderer.h"

namespace uhip {

// ncclReduce(sum) of ctx's W x H x 3 device sum into `device_buffer` of rank `root` (every rank passes a device
// buffer of W*H*3 doubles on ctx's device; on return root's holds the total). `nccl_comm` is an ncclComm_t,
// `hip_stream` a hipStream_t (nullptr: the default stream); the call returns when the reduce has completed.
// Returns a UBDPT_* status (UBDPT_HIP_ERROR for RCCL and HIP failures).
int reduceSumRccl(ubdpt_context* ctx, void* nccl_comm, int root, double* device_buffer, void* hip_stream);

class MultiGpuRenderer {
public:
    // One context per entry of `devices`; a device may be listed more than once (its contexts then share it).
    explicit MultiGpuRenderer(std::vector<int> devices, const ubdpt_options* options = nullptr);
    ~MultiGpuRenderer();
    MultiGpuRenderer(const MultiGpuRenderer&) = delete;
    MultiGpuRenderer& operator=(const MultiGpuRenderer&) = delete;

    bool initialize(const RenderParameters& params, const ubdpt_scene_desc& scene);
    // Passes [first_pass, first_pass + n): context k takes first_pass + k, + K, ... All contexts render at once.
    bool renderPasses(size_t first_pass, size_t n, uint64_t seed, size_t* done = nullptr);
    // Sum of all contexts' device sums -> out ([y][x][rgb]); through RCCL when more than one device is involved,
    // and through the same ncclReduce call (a communicator of one rank) when not.
    bool reduceSum(SumBuffer& out);
    bool reduceSumTo(double* host_rgb);   // the same into W * H * 3 doubles of the caller: one copy, device 0 -> host
    // The reduced sum left on the first device of the list ([y][x][rgb], W * H * 3 doubles, valid until the next
    // reduce): for a preview every few passes nothing has to cross PCIe but what the caller then makes of it.
    // nullptr on failure. The first device is current on return.
    const double* reduceSumOnDevice();
    void clear();  // zeroes every context's device sum
    void stop();
    size_t contexts() const { return m_ctx.size(); }
    int lastStatus() const { return m_status; }

private:
    void release();
    std::vector<int> m_devices;        // per context
    std::vector<int> m_unique;         // distinct devices, communicator rank order
    ubdpt_options m_options{};
    std::vector<ubdpt_context*> m_ctx;
    std::vector<void*> m_comms;        // ncclComm_t per distinct device
    std::vector<double*> m_partial;    // per distinct device: W*H*3 doubles
    size_t m_width = 0, m_height = 0;
    int m_status = UBDPT_SUCCESS;
};

}  // namespace uhip
