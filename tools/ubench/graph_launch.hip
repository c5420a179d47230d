// Micro-benchmark: host time of hipGraphLaunch of an N-kernel-node graph against N hipLaunchKernelGGL calls (gfx950, ROCm 7).
// Kernels take a 1 KB by-value argument block like the binning kernels of the library.  Build: hipcc --offload-arch=gfx950 -O3
// graph_launch.hip -o graph_launch.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { unsigned long long p[128]; };
__global__ void k(Args a, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && a.p[5] == 77ull) out[0] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* out; hipMalloc(&out, 64);
  Args a{}; 
  hipStream_t s; hipStreamCreate(&s);
  for (int N : {1, 8, 28, 45}) {
    // eager
    for (int rep = 0; rep < 3; rep++) for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, a, out);
    hipStreamSynchronize(s);
    const int R = 200;
    double t0 = now();
    for (int r = 0; r < R; r++) { for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, a, out); }
    double t1 = now();
    hipStreamSynchronize(s);
    double t1b = now();
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, a, out);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; rep++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double t2 = now();
    for (int r = 0; r < R; r++) hipGraphLaunch(ge, s);
    double t3 = now();
    hipStreamSynchronize(s);
    double t3b = now();
    // launch into the legacy default stream
    double t4 = now();
    for (int r = 0; r < R; r++) hipGraphLaunch(ge, 0);
    double t5 = now();
    hipDeviceSynchronize();
    printf("N=%2d  eager %.1f us per sequence host (%.2f per launch; drained %.1f)   graph %.1f us per launch host (%.2f per node; drained %.1f)   graph on the null stream %.1f us\n",
           N, (t1 - t0) / R * 1e6, (t1 - t0) / R / N * 1e6, (t1b - t0) / R * 1e6, (t3 - t2) / R * 1e6, (t3 - t2) / R / N * 1e6, (t3b - t2) / R * 1e6, (t5 - t4) / R * 1e6);
    // cost of making a graph: capture, instantiate, and updating an instantiated one from a fresh capture
    {
      const int R2 = 50;
      double tc = 0, ti = 0, tu = 0, td = 0, tg = 0;
      for (int r = 0; r < R2; r++) {
        a.p[7] = (unsigned long long)r;
        double c0 = now();
        hipGraph_t g2; hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, a, out);
        hipStreamEndCapture(s, &g2);
        double c1 = now();
        hipGraphExec_t ge2; hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        double c2 = now();
        hipGraphNode_t en; hipGraphExecUpdateResult ur;
        hipError_t ue = hipGraphExecUpdate(ge, g2, &en, &ur);
        double c3 = now();
        if (r == 0) printf("      (hipGraphExecUpdate: %s, result %d)\n", hipGetErrorString(ue), (int)ur);
        hipGraphExecDestroy(ge2);
        double c3b = now();
        hipGraphDestroy(g2);
        double c4 = now();
        tc += c1 - c0; ti += c2 - c1; tu += c3 - c2; td += c3b - c3; tg += c4 - c3b;
      }
      printf("      capture %.1f us, instantiate %.1f us, exec-update %.1f us, exec destroy %.1f us, graph destroy %.1f us (N=%d)\n", tc / R2 * 1e6, ti / R2 * 1e6, tu / R2 * 1e6, td / R2 * 1e6, tg / R2 * 1e6, N);
      hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
