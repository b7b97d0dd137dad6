// How many bytes does HBM hand out for ONE looked-up 4-byte word when every lookup lands in a
// line of its own?  Two kernels over the same random indices into a 1 GiB table:
//   vector: lane l loads table[idx[g]] with a per-lane (vector) load -- what the round kernels do
//   scalar: the wave walks its 64 indices with readlane and loads each word with a scalar load
//           (uniform address), 16 loads in flight at a time
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum
// TCC_EA0_RDREQ_128B_sum to see the request sizes each produces; prints the time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(128) void gather_vector(const int *__restrict__ table, const uint32_t *__restrict__ idx,
                                                      int *__restrict__ out, uint32_t G) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    out[g] = table[idx[g]];
}

template <int MODE>
__global__ __launch_bounds__(128) void gather_flavour(const int *__restrict__ table, const uint32_t *__restrict__ idx,
                                                      int *__restrict__ out, uint32_t G) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int *p = table + idx[g];
    int v;
    if (MODE == 0) v = __builtin_nontemporal_load(p);
    else if (MODE == 1) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else v = (int)__hip_atomic_load(reinterpret_cast<const unsigned char *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    out[g] = v;
}

__global__ __launch_bounds__(128) void gather_scalar(const int *__restrict__ table, const uint32_t *__restrict__ idx,
                                                      int *__restrict__ out, uint32_t G) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t my = g < G ? idx[g] : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    int mine = 0;
    for (int c = 0; c < 64; c += 16) {
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t i = __builtin_amdgcn_readlane(my, c + k);     // uniform: a scalar load follows
            v[k] = table[i];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) mine = lane == (uint32_t)(c + k) ? v[k] : mine;
    }
    if (g < G) out[g] = mine;
}

int main() {
    const uint32_t G = 1u << 20;
    const size_t words = 1ull << 28;                  // 1 GiB table: every lookup in a line of its own
    int *table; uint32_t *idx; int *out;
    CK(hipMalloc(&table, words * 4)); CK(hipMemset(table, 1, words * 4));
    std::vector<uint32_t> h(G * 8);
    uint64_t s = 12345;
    for (auto &x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (uint32_t)(s >> 36); }
    CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, G * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 6; mode++) {
        const int iters = 40;
        for (int i = -4; i < iters; i++) {
            if (i == 0) CK(hipEventRecord(e0));
            const uint32_t *ix = idx + (size_t)((i + 8) % 8) * G;      // fresh indices each launch
            if (mode == 0) hipLaunchKernelGGL(gather_vector, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
            else if (mode == 1) hipLaunchKernelGGL(gather_scalar, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
            else if (mode == 2) hipLaunchKernelGGL(gather_flavour<0>, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
            else if (mode == 3) hipLaunchKernelGGL(gather_flavour<1>, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
            else if (mode == 4) hipLaunchKernelGGL(gather_flavour<2>, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
            else hipLaunchKernelGGL(gather_flavour<3>, dim3(G / 128), dim3(128), 0, 0, table, ix, out, G);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char *names[] = {"vector", "scalar", "vector nt", "vector agent-scope", "vector system-scope", "vector system-scope byte"};
        printf("%s gather of %u words: %.2f us per launch\n", names[mode], G, ms * 1e3 / iters);
    }
    return 0;
}
