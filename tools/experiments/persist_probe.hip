// Probe behind DESIGN.md's persistent-kernel worksheet ("build the persistent / cooperative 19-step
// kernel or prove it loses").  Measures, at THIS path's launch geometry (256 workgroups, one per CU, 1024 or 256 threads),
// what the seams of a fused recurrent step would cost against what the kernel boundaries they replace cost:
//   chain     K dependent trivial launches                                  -> us per kernel boundary
//   grid      ONE launch, K rounds of {publish P bytes per workgroup with write-through (sc1) stores, grid barrier}
//             flat counter and XCD-hierarchical barrier                      -> us per in-launch all-to-all seam
//   cluster   ONE launch, 32 clusters of 8 workgroups (one per XCD: blocks 8c .. 8c+7), K rounds of {publish 32 KB, flag,
//             wait for the 7 partners' flags, read their 8 x 32 KB}          -> us per scene-local exchange (the y1 tile of a
//             64-ego tile split over 8 column-block workgroups)
// Every spin is bounded (a give-up code is reported instead of a hang).  Nothing here is linked into libtrajnet_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/persist_probe tools/experiments/persist_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_sc1(f4 *p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f4 load_sc1(const f4 *p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned poll(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- chain: a trivial dependent kernel -----------------------------------------------------------------------
__global__ void chain_kernel(const float *in, float *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = in[(blockIdx.x + 1) % gridDim.x] + 1.0f;
}

// ---- grid barrier forms ----------------------------------------------------------------------------------------
struct Bar { unsigned *flat; unsigned *xcc; unsigned *top; unsigned *gen; int *giveup; };

__device__ __forceinline__ bool spin_until(const unsigned *p, unsigned target, int *giveup) {
    for (int it = 0; it < (1 << 22); ++it) {
        if ((int)(poll(p) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    *giveup = 1;
    return false;
}

// flat: one monotonic counter, lane 0 of every workgroup arrives and polls
__device__ void barrier_flat(const Bar &b, unsigned round, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(b.flat, round * nblocks, b.giveup);
    }
    __syncthreads();
}

// XCD-hierarchical: arrivals counted per XCD (block b runs on XCD b % 8), the last arriver of an XCD adds to the top
// counter, the last XCD bumps a generation word per XCD that the members poll
__device__ void barrier_xcd(const Bar &b, unsigned round, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7u, per = nblocks / 8u;
        const unsigned prev = __hip_atomic_fetch_add(b.xcc + 32 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == round * per) {                                // last of this XCD
            const unsigned pt = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pt + 1 == round * 8u) {                               // last XCD: release everybody
                for (int k = 0; k < 8; ++k) __hip_atomic_store(b.gen + 32 * k, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        spin_until(b.gen + 32 * x, round, b.giveup);
    }
    __syncthreads();
}

template <int FORM>
__global__ void __launch_bounds__(1024) grid_rounds_kernel(Bar b, f4 *slab, int payload_f4_per_block, int rounds, float *sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 *mine = slab + (size_t)blockIdx.x * payload_f4_per_block;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = threadIdx.x; i < payload_f4_per_block; i += blockDim.x) {
            f4 v = {(float)r, acc.x, 1.f, 2.f};
            store_sc1(mine + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FORM == 0) barrier_flat(b, (unsigned)r, gridDim.x);
        else barrier_xcd(b, (unsigned)r, gridDim.x);
        // read a neighbour's slab head (what a consumer phase would do first)
        if (payload_f4_per_block > 0 && threadIdx.x < 64) {
            const f4 *other = slab + (size_t)((blockIdx.x + 9) % gridDim.x) * payload_f4_per_block;
            f4 v = load_sc1(other + threadIdx.x % payload_f4_per_block);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
            acc.x += v.x;
        }
    }
    if (acc.x == 1.2345e30f) sink[0] = acc.x;
}

// ---- cluster exchange ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) cluster_rounds_kernel(f4 *slab, unsigned *flags, int f4_per_block, int rounds, int *giveup,
                                                              float *sink, int same_xcd) {
    // cluster = 8 workgroups.  same_xcd == 0: blocks 8c .. 8c+7 (one per XCD); 1: blocks c + 32 k, k = 0..7 -> all on XCD c % 8
    int cl, member;
    if (!same_xcd) { cl = blockIdx.x >> 3; member = blockIdx.x & 7; }
    else { cl = blockIdx.x & 31; member = blockIdx.x >> 5; }
    f4 *base = slab + (size_t)cl * 8 * f4_per_block;
    unsigned *fl = flags + (size_t)cl * 8 * 32;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 1; r <= rounds; ++r) {
        f4 *mine = base + (size_t)member * f4_per_block;
        for (int i = threadIdx.x; i < f4_per_block; i += blockDim.x) {
            f4 v = {(float)r, acc.y, (float)member, 3.f};
            store_sc1(mine + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(fl + 32 * member, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 8) spin_until(fl + 32 * threadIdx.x, (unsigned)r, giveup);
        __syncthreads();
        // every member reads the whole 8-slab tile (the K = 1024 operand of the second layer), 8 loads in flight per lane
        const int tot = 8 * f4_per_block;
        for (int i0 = threadIdx.x; i0 < tot; i0 += blockDim.x * 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x; v[u] = load_sc1(base + (i < tot ? i : 0)); }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc.y += v[u].x * 1e-9f;
        }
        __syncthreads();
    }
    if (acc.y == 1.2345e30f) sink[0] = acc.y;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0.f; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NB = 256;
    float *bufa, *bufb, *sink;
    CK(hipMalloc(&bufa, NB * 4)); CK(hipMalloc(&bufb, NB * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(bufa, 0, NB * 4)); CK(hipMemset(bufb, 0, NB * 4));
    printf("persist_probe: 256 workgroups (one per CU); all times are means over the rounds of a run, best of 3 runs\n\n");
    // ---- chain ----
    for (int threads : {256, 1024}) {
        const int K = 400;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(chain_kernel, dim3(NB), dim3(threads), 0, 0, bufa, bufb);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(chain_kernel, dim3(NB), dim3(threads), 0, 0, (i & 1) ? bufb : bufa, (i & 1) ? bufa : bufb);
            CK(hipEventRecord(e1));
            const float ms = time_ms(e0, e1);
            best = ms < best ? ms : best;
        }
        printf("chain     %4d threads/WG: %6.2f us per dependent trivial launch (boundary + the kernel itself)\n", threads, best * 1e3f / K);
    }
    // ---- grid barrier ----
    unsigned *ctr;
    int *giveup;
    CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&giveup, 4));
    const size_t slab_bytes = (size_t)NB * 64 * 1024;
    f4 *slab;
    CK(hipMalloc(&slab, slab_bytes));
    for (int form = 0; form < 2; ++form)
        for (int payload_kb : {0, 4, 32, 64}) {
            const int rounds = 200;
            float best = 1e9f;
            int gu = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctr, 0, 4096 * 4)); CK(hipMemset(giveup, 0, 4));
                Bar b = {ctr, ctr + 64, ctr + 1024, ctr + 2048, giveup};
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(grid_rounds_kernel<0>, dim3(NB), dim3(1024), 0, 0, b, slab, payload_kb * 64, rounds, sink);
                else hipLaunchKernelGGL(grid_rounds_kernel<1>, dim3(NB), dim3(1024), 0, 0, b, slab, payload_kb * 64, rounds, sink);
                CK(hipEventRecord(e1));
                const float ms = time_ms(e0, e1);
                best = ms < best ? ms : best;
                CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost));
            }
            printf("grid      %-12s barrier, %2d KB published per WG per round (sc1 stores): %6.2f us per round%s\n",
                   form == 0 ? "flat-counter" : "xcd-hier.", payload_kb, best * 1e3f / rounds, gu ? "  [GAVE UP: a spin timed out]" : "");
        }
    // ---- cluster exchange ----
    unsigned *flags;
    CK(hipMalloc(&flags, 32 * 8 * 32 * 4));
    for (int same = 0; same < 2; ++same)
        for (int kb : {8, 32}) {
            const int rounds = 200;
            float best = 1e9f;
            int gu = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(flags, 0, 32 * 8 * 32 * 4)); CK(hipMemset(giveup, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(cluster_rounds_kernel, dim3(NB), dim3(1024), 0, 0, slab, flags, kb * 64, rounds, giveup, sink, same);
                CK(hipEventRecord(e1));
                const float ms = time_ms(e0, e1);
                best = ms < best ? ms : best;
                CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost));
            }
            printf("cluster   8 WGs %-22s publish %2d KB each + flags + read all 8 slabs (%3d KB): %6.2f us per round%s\n",
                   same ? "on one XCD" : "spread over the 8 XCDs", kb, 8 * kb, best * 1e3f / rounds, gu ? "  [GAVE UP]" : "");
        }
    printf("\n(kernel boundary for comparison: the chain rows above minus ~1 us of kernel body; the product's recurrent step has four\n"
           " boundaries and, fused, would have three to four of the seams measured here)\n");
    return 0;
}
