// What a launch costs before and after its memory traffic: kernels with the fused round's grid
// (one thread per group) that move nothing, one 4-byte stream in and out, or one dependent chain
// of the round's shape (load -> indexed load -> store), timed back to back on one stream.
//   hipcc -O3 --offload-arch=gfx950 tools/launch_probe.hip -o tools/launch_probe && tools/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__global__ void k_empty(const int *in, int *out, unsigned G) {
    unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0xFFFFFFFFu) out[0] = in[0];
}
__global__ void k_copy(const int *in, int *out, unsigned G) {
    unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) __builtin_nontemporal_store(__builtin_nontemporal_load(in + g), out + g);
}
__global__ void k_chain(const int *in, int *out, unsigned G) {
    unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int a = __builtin_nontemporal_load(in + g);
    int b = in[(unsigned)(a + g) % G];          // dependent, same neighbourhood
    __builtin_nontemporal_store(a + b, out + g);
}

template <typename K>
static double run(K k, unsigned G, unsigned block, const int *in, int *out, int launches) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned grid = (G + block - 1) / block;
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, in, out, G);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, in, out, G);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / launches;
}

// the same launches captured into one HIP graph and replayed
template <typename K>
static double run_graph(K k, unsigned G, unsigned block, const int *in, int *out, int launches) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned grid = (G + block - 1) / block;
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, s, in, out, G);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(s));
    return ms * 1e3 / launches;
}

int main() {
    const unsigned GMAX = 8u << 20;
    int *in, *out;
    CK(hipMalloc(&in, (size_t)GMAX * 4)); CK(hipMalloc(&out, (size_t)GMAX * 4));
    CK(hipMemset(in, 0, (size_t)GMAX * 4));
    printf("%-8s %9s %6s %10s\n", "kernel", "groups", "block", "us/launch");
    for (unsigned G : {10000u, 1u << 18, 1u << 20, 4u << 20, 8u << 20})
        for (unsigned block : {64u, 128u, 256u}) {
            printf("%-8s %9u %6u %10.2f\n", "empty", G, block, run(k_empty, G, block, in, out, 400));
            printf("%-8s %9u %6u %10.2f\n", "copy", G, block, run(k_copy, G, block, in, out, 400));
            printf("%-8s %9u %6u %10.2f\n", "chain", G, block, run(k_chain, G, block, in, out, 400));
        }
    printf("\n%-8s %9s %6s %10s %10s\n", "kernel", "groups", "block", "eager us", "graph us");
    for (unsigned G : {10000u, 1u << 20, 4u << 20, 8u << 20}) {
        printf("%-8s %9u %6u %10.2f %10.2f\n", "copy", G, 256u, run(k_copy, G, 256, in, out, 400), run_graph(k_copy, G, 256, in, out, 400));
        printf("%-8s %9u %6u %10.2f %10.2f\n", "chain", G, 256u, run(k_chain, G, 256, in, out, 400), run_graph(k_chain, G, 256, in, out, 400));
    }
    return 0;
}
