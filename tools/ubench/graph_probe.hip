// Which stream-capture pattern does the HIP runtime on the MI355X box take?  (round 6: hipStreamEndCapture crashed on the frame's three-stream
// fork / join capture.)   hipcc --offload-arch=gfx950 -O2 tools/ubench/graph_probe.hip -o tools/ubench/graph_probe ; ./graph_probe <variant>
//   1 one stream: memset + kernels            2 fork / join with ONE side stream            3 two side streams with cross dependencies (the frame's shape)
//   4 = 3 with priority side streams          5 = 3 in global capture mode                  6 = 3 with an event recorded twice
//   7 = 3 where a side stream's LAST captured operation is an event record nobody waits for (plus the proper join before it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorName(e_)); fflush(stdout); } } while (0)
__global__ void k(float* p, int n, float v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 0.5f + v; }
int main(int argc, char** argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 1;
    printf("variant %d\n", var); fflush(stdout);
    float* buf; const int n = 1 << 20;
    CK(hipMalloc(&buf, 4 * n * sizeof(float)));
    hipStream_t cap, s1, s2;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    if (var == 4) { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi)); }
    else { CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
    hipEvent_t ev[8];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // some eager work first (the frame's first sighting is rendered eagerly on the same streams and events)
    CK(hipEventRecord(ev[0], cap)); CK(hipStreamWaitEvent(s1, ev[0], 0)); k<<<n / 256, 256, 0, s1>>>(buf, n, 1.f); CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(cap, ev[1], 0));
    CK(hipDeviceSynchronize());
    CK(hipStreamBeginCapture(cap, var == 5 ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(buf, 0, n * sizeof(float), cap));
    k<<<n / 256, 256, 0, cap>>>(buf, n, 1.f);
    if (var >= 2) {
        CK(hipEventRecord(ev[0], cap));
        CK(hipStreamWaitEvent(s1, ev[0], 0));
        if (var >= 3) CK(hipStreamWaitEvent(s2, ev[0], 0));
        CK(hipMemsetAsync(buf + n, 0, n * sizeof(float), s1));
        k<<<n / 256, 256, 0, s1>>>(buf + n, n, 2.f);
        if (var >= 3) {
            CK(hipEventRecord(ev[2], s1)); CK(hipStreamWaitEvent(s2, ev[2], 0));            // s2 after s1's first part
            k<<<n / 256, 256, 0, s2>>>(buf + 2 * n, n, 3.f);
            CK(hipEventRecord(ev[3], s2));                                                    // s2 -> main (like ev_smpl)
            k<<<n / 256, 256, 0, s2>>>(buf + 2 * n, n, 4.f);
            CK(hipEventRecord(ev[4], s2)); CK(hipStreamWaitEvent(s1, ev[4], 0));            // s1 after s2's second part (level builds)
            if (var == 6) { CK(hipEventRecord(ev[2], s1)); CK(hipStreamWaitEvent(s2, ev[2], 0)); }
            k<<<n / 256, 256, 0, s1>>>(buf + n, n, 5.f);
            CK(hipEventRecord(ev[5], s1)); CK(hipStreamWaitEvent(s2, ev[5], 0));            // s2 after s1 (folds)
            k<<<n / 256, 256, 0, s2>>>(buf + 3 * n, n, 6.f);
            CK(hipEventRecord(ev[6], s2)); CK(hipStreamWaitEvent(s1, ev[6], 0));            // s2 joins s1
            if (var == 7) CK(hipEventRecord(ev[7], s2));
        }
        k<<<n / 256, 256, 0, cap>>>(buf, n, 7.f);                                             // main's own chain meanwhile
        CK(hipEventRecord(ev[1], s1));
        if (var >= 3) CK(hipStreamWaitEvent(cap, ev[3], 0));
        CK(hipStreamWaitEvent(cap, ev[1], 0));                                                // s1 (and through it s2) joins main
    }
    k<<<n / 256, 256, 0, cap>>>(buf, n, 8.f);
    hipGraph_t g = nullptr; hipGraphExec_t x = nullptr;
    printf("  end capture...\n"); fflush(stdout);
    CK(hipStreamEndCapture(cap, &g));
    printf("  ended, graph %p\n", (void*)g); fflush(stdout);
    if (g) { CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0)); printf("  instantiated %p\n", (void*)x); fflush(stdout); }
    if (x) { for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, cap)); CK(hipStreamSynchronize(cap)); CK(hipGraphLaunch(x, nullptr)); CK(hipDeviceSynchronize()); }
    float h[4]; for (int i = 0; i < 4; ++i) CK(hipMemcpy(&h[i], buf + (size_t)i * n + 5, sizeof(float), hipMemcpyDeviceToHost));
    printf("  values %g %g %g %g: OK\n", h[0], h[1], h[2], h[3]);
    return 0;
}
