// Torch-free reproducer attempt for the ROCm 7.2 hipGraph replay defect described in DESIGN.md 5c: a training step captured as a
// CHAIN of graphs (stream capture, one hipGraphExec per segment) and relaunched back to back WITHOUT a stream synchronisation in
// between produced NaN gradients from the second replay on unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.
//
// What the captured step looks like, reduced to its node kinds (everything in integer arithmetic: the reference is the SAME node sequence
// enqueued eagerly, compared word for word):
//   * kernel nodes with LARGE by-value argument structs (GemmParams is ~400 bytes) and many of them per graph,
//   * hipMemsetAsync nodes (bdetr_zero; the strided backward-data memset),
//   * float-atomic-like accumulation into a buffer zeroed by such a memset node,
//   * a kernel that reads per-step scalars from device memory (dropout seed, learning rate) that an EAGER fill kernel rewrites
//     between two replays,
//   * an eager device-to-device copy of the batch into the graph's static input between replays,
//   * a last kernel that writes a word to a PINNED host ring (the range guard's snapshot),
//   * optional: side graphs launched on a second stream behind an event (argv "side").
// The chain is replayed `steps` times (a) with a hipStreamSynchronize after every step and (b) with none; both must end in the
// state of the eager reference.
//
// Build: hipcc --offload-arch=gfx950 -O2 graph_replay_repro.hip -o graph_replay_repro
// Run  : ./graph_replay_repro [segments=16] [kernels_per_segment=40] [steps=12] [side]
//        with DEBUG_CLR_GRAPH_PACKET_CAPTURE unset (runtime default) and =0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct BigArgs {                 // ~416 bytes by value, like bdgemm::GemmParams
    uint32_t seg, k, n, m;
    uint32_t* state; uint32_t* scratch; const uint32_t* scalars; const uint32_t* input;
    uint32_t pad[88];
    uint32_t salt;
};

constexpr int N = 1 << 20;
static int M = 4096;             // words zeroed by the memset NODE of every segment (argv[5]; the failing one of the framework is 409600 = 1.6 MB)

__host__ __device__ inline uint32_t mix(uint32_t s, uint32_t seed, uint32_t seg, uint32_t k, uint32_t salt, uint32_t in) {
    return s * 1664525u + 1013904223u + seed * 2654435761u + seg * 97u + k * 13u + salt + in;
}

__global__ void step_kernel(BigArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    a.state[i] = mix(a.state[i], a.scalars[0], a.seg, a.k, a.salt ^ a.pad[(a.k * 7) % 88], a.input[i % a.m]);
}
// "split-K atomics": scratch (zeroed by a memset NODE in the same graph) += state, integer atomics (order independent)
__global__ void accum_kernel(BigArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    atomicAdd(&a.scratch[i % a.m], a.state[i]);
}
__global__ void fold_kernel(BigArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    a.state[i] ^= a.scratch[i % a.m] + a.scalars[1];
}
// the memset's target is pool memory that another tensor of the step used before: poison it first, so that a memset node that runs
// out of order (or not at all) shows
__global__ void poison_kernel(uint32_t* p, int m) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < m) p[i] = 0x7FC00000u + (uint32_t)i; }
__global__ void fill_kernel(uint32_t* p, uint32_t v) { *p = v; }                        // torch's fill_ of a device scalar
__global__ void snapshot_kernel(const uint32_t* state, uint32_t* ordinal, uint32_t* ring, int ring_len) {
    const uint32_t k = ++*ordinal;
    __hip_atomic_store(&ring[1 + k % ring_len], state[k % 1024], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&ring[0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static BigArgs make_args(int seg, int k, uint32_t* state, uint32_t* scratch, const uint32_t* scalars, const uint32_t* input) {
    BigArgs a;
    memset(&a, 0, sizeof a);
    a.seg = seg; a.k = k; a.n = N; a.m = M; a.state = state; a.scratch = scratch; a.scalars = scalars; a.input = input;
    for (int j = 0; j < 88; ++j) a.pad[j] = (uint32_t)(seg * 1000 + k * 31 + j);
    a.salt = (uint32_t)(seg * 7919 + k);
    return a;
}

int main(int argc, char** argv) {
    const int segments = argc > 1 ? atoi(argv[1]) : 16, kper = argc > 2 ? atoi(argv[2]) : 40, steps = argc > 3 ? atoi(argv[3]) : 12;
    const bool side = argc > 4 && !strcmp(argv[4], "side");
    if (argc > 5) M = atoi(argv[5]);
    const int sync_steps = argc > 6 ? atoi(argv[6]) : 0;          // pass 2: synchronise after each of the first sync_steps steps only
    const char* env = getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
    printf("graph_replay_repro: %d segments x %d kernels, %d steps, side=%d, memset words %d, sync_steps %d, DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s\n", segments, kper, steps, (int)side, M, sync_steps, env ? env : "(unset)");
    uint32_t *state, *scratch, *scratch2, *scalars, *input, *batch[2], *ordinal, *ring;
    CK(hipMalloc(&state, N * 4)); CK(hipMalloc(&scratch, M * 4)); CK(hipMalloc(&scratch2, M * 4)); CK(hipMalloc(&scalars, 8)); CK(hipMalloc(&input, M * 4));
    CK(hipMalloc(&batch[0], M * 4)); CK(hipMalloc(&batch[1], M * 4)); CK(hipMalloc(&ordinal, 4));
    CK(hipHostMalloc(&ring, 9 * 4, hipHostMallocDefault));
    std::vector<uint32_t> hb[2] = {std::vector<uint32_t>(M), std::vector<uint32_t>(M)};
    for (int i = 0; i < M; ++i) { hb[0][i] = i * 2654435761u; hb[1][i] = ~(i * 40503u); }
    CK(hipMemcpy(batch[0], hb[0].data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(batch[1], hb[1].data(), M * 4, hipMemcpyHostToDevice));
    hipStream_t cap, mainS, sideS;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&sideS, hipStreamNonBlocking, lo));

    // ---- capture: main segments (and, with "side", a side graph per segment that works on scratch2 only) ----
    std::vector<hipGraphExec_t> mains, sides;
    for (int s = 0; s < segments; ++s) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(poison_kernel, dim3((M + 255) / 256), dim3(256), 0, cap, scratch, M);
        CK(hipMemsetAsync(scratch, 0, M * 4, cap));
        for (int k = 0; k < kper; ++k) {
            BigArgs a = make_args(s, k, state, scratch, scalars, input);
            hipLaunchKernelGGL(step_kernel, dim3(N / 256), dim3(256), 0, cap, a);
            if (k % 8 == 3) hipLaunchKernelGGL(accum_kernel, dim3(N / 256), dim3(256), 0, cap, a);
        }
        hipLaunchKernelGGL(fold_kernel, dim3(N / 256), dim3(256), 0, cap, make_args(s, 0, state, scratch, scalars, input));
        if (s == segments - 1) hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, cap, state, ordinal, ring, 8);
        CK(hipStreamEndCapture(cap, &g));
        hipGraphExec_t e; CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
        mains.push_back(e);
        if (side) {
            CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(scratch2, 0, M * 4, cap));
            for (int k = 0; k < 6; ++k) {
                BigArgs a = make_args(s, k, state, scratch2, scalars, input);
                a.n = 0;                                   // (reads nothing, writes nothing: the side work must not race with the main chain's state)
                hipLaunchKernelGGL(accum_kernel, dim3(N / 256), dim3(256), 0, cap, a);
            }
            CK(hipStreamEndCapture(cap, &g));
            CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
            sides.push_back(e);
        }
    }

    // the same node sequence enqueued eagerly (no graphs): the reference
    auto segment_eager = [&](int sgm, hipStream_t st, bool last) {
        hipLaunchKernelGGL(poison_kernel, dim3((M + 255) / 256), dim3(256), 0, st, scratch, M);
        CK(hipMemsetAsync(scratch, 0, M * 4, st));
        for (int k = 0; k < kper; ++k) {
            BigArgs a = make_args(sgm, k, state, scratch, scalars, input);
            hipLaunchKernelGGL(step_kernel, dim3(N / 256), dim3(256), 0, st, a);
            if (k % 8 == 3) hipLaunchKernelGGL(accum_kernel, dim3(N / 256), dim3(256), 0, st, a);
        }
        hipLaunchKernelGGL(fold_kernel, dim3(N / 256), dim3(256), 0, st, make_args(sgm, 0, state, scratch, scalars, input));
        if (last) hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, st, state, ordinal, ring, 8);
    };
    std::vector<uint32_t> init(N), want(N), got(N);
    for (int i = 0; i < N; ++i) init[i] = (uint32_t)i * 3u + 1u;
    int bad_total = 0;
    for (int pass = 0; pass < 3; ++pass) {                 // 0: eager reference, 1: graphs, synchronise after every step, 2: graphs, never
        CK(hipMemcpy(state, init.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(ordinal, 0, 4)); memset(ring, 0, 9 * 4);
        CK(hipDeviceSynchronize());
        for (int t = 0; t < steps; ++t) {
            // eager work between replays, as Model._graph_step does it: copy the batch into the static input, fill the two scalars
            CK(hipMemcpyAsync(input, batch[t & 1], M * 4, hipMemcpyDeviceToDevice, mainS));
            hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(1), 0, mainS, scalars, 0x5EEDu + t);
            hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(1), 0, mainS, scalars + 1, 77u * t + 1u);
            bool used = false;
            for (int sgm = 0; sgm < segments; ++sgm) {
                if (pass == 0) { segment_eager(sgm, mainS, sgm == segments - 1); continue; }
                if (sgm == segments - 1 && used) { hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventRecord(ev, sideS)); CK(hipStreamWaitEvent(mainS, ev, 0)); CK(hipEventDestroy(ev)); }
                CK(hipGraphLaunch(mains[sgm], mainS));
                if (side) {
                    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventRecord(ev, mainS)); CK(hipStreamWaitEvent(sideS, ev, 0)); CK(hipEventDestroy(ev));
                    CK(hipGraphLaunch(sides[sgm], sideS));
                    used = true;
                }
            }
            if (pass == 1 || (pass == 2 && t < sync_steps)) CK(hipDeviceSynchronize());
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(pass == 0 ? want.data() : got.data(), state, N * 4, hipMemcpyDeviceToHost));
        if (pass == 0) { printf("  pass 0 (eager reference): ring ordinal %u\n", ring[0]); continue; }
        int bad = 0, first = -1;
        for (int i = 0; i < N; ++i) if (got[i] != want[i]) { if (first < 0) first = i; ++bad; }
        printf("  pass %d (%s): %d of %d words differ from the eager reference (first at %d), ring ordinal %u\n", pass, pass == 1 ? "graphs, sync after every step" : "graphs, no sync between steps", bad, N, first, ring[0]);
        bad_total += bad;
    }
    printf(bad_total ? "GRAPH_REPLAY_REPRO: MISMATCH\n" : "GRAPH_REPLAY_REPRO: CLEAN\n");
    return 0;
}
