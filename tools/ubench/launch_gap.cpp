// Device-side cost of a kernel boundary on one stream (MI355X, 8 XCDs): period of dependent tiny kernels launched (a) one by one,
// (b) with hipExtAnyOrderLaunch (no barrier between them), (c) as a captured hipGraph replayed.  Decides whether the tracker's
// 2 launches x 20 iterations and the single-pair (step, finalize) pair should be graph launches.
//   hipcc -O3 --offload-arch=gfx950 launch_gap.cpp -o launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_chain(float* p, int work) {   // reads what the previous kernel wrote, `work` dependent fmas per thread
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < work; ++i) v = v * 1.0000001f + 1e-9f;
  p[threadIdx.x & 63] = v;
}

static double time_ms(hipStream_t s, hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); (void)s; return ms; }

int main() {
  float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 2000;
  for (int grid : {1, 256, 4096}) for (int work : {0, 2000}) {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, p, work);
    CK(hipEventRecord(a, s));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, p, work);
    CK(hipEventRecord(b, s));
    const double t_plain = time_ms(s, a, b) / N * 1e3;
    CK(hipEventRecord(a, s));
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p, work);
    CK(hipEventRecord(b, s));
    const double t_any = time_ms(s, a, b) / N * 1e3;
    // graph: 40 dependent kernel nodes captured from the stream, replayed
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, p, work);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    const double t_graph = time_ms(s, a, b) / (50 * 40) * 1e3;
    // one blocking (launch pair + sync) call, as the single-pair operators do
    CK(hipStreamSynchronize(s));
    hipEvent_t c0, c1; CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    CK(hipEventRecord(c0, s));
    for (int i = 0; i < 300; ++i) { hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, p, work); hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, p, work); CK(hipStreamSynchronize(s)); }
    CK(hipEventRecord(c1, s));
    const double t_block = time_ms(s, c0, c1) / 300 * 1e3;
    printf("grid %4d work %4d: period per kernel  plain %6.2f us   any-order %6.2f us   graph node %6.2f us   | blocking (2 kernels + sync) %6.2f us\n", grid, work, t_plain, t_any, t_graph, t_block);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
