// Streaming ceiling probe: a kernel with the fused round's traffic shape (bytes read : bytes
// written = 100 : 34 per element, 4-byte accesses, 128-thread blocks, non-temporal) but a
// configurable number of concurrent streams, to separate "many planes" from "HBM mix".
//   layout 0: K read planes + M write planes, plane-major (field[p][g])  -- what librq uses
//   layout 1: the same words tiled per 64 groups (tile[g/64][field][g%64]) -- one stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int K, int M, bool TILED>
__global__ __launch_bounds__(128) void probe(const int* __restrict__ in, int* __restrict__ out, size_t G, size_t ld) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int acc = 0;
    int v[K > 0 ? K : 1];
#pragma unroll
    for (int p = 0; p < K; p++) {
        size_t idx = TILED ? ((g >> 6) * (size_t)K * 64 + (size_t)p * 64 + (g & 63)) : (size_t)p * ld + g;
        v[p] = __builtin_nontemporal_load(in + idx);
    }
#pragma unroll
    for (int p = 0; p < K; p++) acc = max(acc, v[p]);
#pragma unroll
    for (int p = 0; p < M; p++) {
        size_t idx = TILED ? ((g >> 6) * (size_t)M * 64 + (size_t)p * 64 + (g & 63)) : (size_t)p * ld + g;
        __builtin_nontemporal_store(acc + p, out + idx);
    }
}

template <int K, int M, bool TILED>
double run(size_t G, int sets, int iters) {
    size_t ld = G + 256;
    std::vector<int*> in(sets), out(sets);
    for (int s = 0; s < sets; s++) {
        CK(hipMalloc(&in[s], ld * K * 4)); CK(hipMalloc(&out[s], ld * M * 4));
        CK(hipMemset(in[s], 1, ld * K * 4)); CK(hipMemset(out[s], 0, ld * M * 4));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((G + 127) / 128), blk(128);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((probe<K, M, TILED>), grid, blk, 0, 0, in[i % sets], out[i % sets], G, ld);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((probe<K, M, TILED>), grid, blk, 0, 0, in[i % sets], out[i % sets], G, ld);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    for (int s = 0; s < sets; s++) { CK(hipFree(in[s])); CK(hipFree(out[s])); }
    double us = ms * 1e3 / iters;
    double bytes = (double)G * (K + M) * 4;
    printf("K=%2d M=%2d %s G=%zu: %.2f us/launch, %.2f TB/s\n", K, M, TILED ? "tiled" : "planes", G, us, bytes / us / 1e6);
    return us;
}

int main() {
    const size_t G = 1 << 20;
    run<25, 9, false>(G, 12, 200);   // the fused round's stream count: 25 x 4 B read, 9 x 4 B written (36 vs 34 B)
    run<25, 9, true>(G, 12, 200);
    run<1, 1, false>(G * 16, 4, 50); // plain copy, same kernel shape
    run<4, 1, false>(G * 8, 4, 50);
    run<25, 0, false>(G, 12, 200);   // reads only
    run<0, 9, false>(G * 2, 12, 200);// writes only
    run<25, 9, false>(G * 8, 2, 30);
    run<25, 9, true>(G * 8, 2, 30);
    return 0;
}
