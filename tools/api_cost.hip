// api_cost.hip -- measurement tool (not part of the product): host-side and
// stream-side cost of the HIP calls the entry points make per call (pointer
// queries, capture query, event record / synchronize, kernel launch), which is
// what decides how the small-call paths are built ("Crossover").
//   build: make -C tools bin/api_cost
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { if (threadIdx.x == 9999) *p = 1; }
int main() {
    (void)hipSetDevice(0);
    void* d; hipMalloc(&d, 1 << 20);
    void* pin; hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault);
    std::vector<char> h(1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 20000;
    hipPointerAttribute_t a;
    double t0 = now();
    for (int i = 0; i < N; ++i) { if (hipPointerGetAttributes(&a, h.data() + (i & 1023) * 8) != hipSuccess) (void)hipGetLastError(); }
    printf("hipPointerGetAttributes(pageable): %.2f us\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; ++i) hipPointerGetAttributes(&a, d);
    printf("hipPointerGetAttributes(device):   %.2f us\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; ++i) hipPointerGetAttributes(&a, pin);
    printf("hipPointerGetAttributes(pinned):   %.2f us\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; ++i) { hipDeviceptr_t b; size_t sz; hipMemGetAddressRange(&b, &sz, d); }
    printf("hipMemGetAddressRange:             %.2f us\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; ++i) { hipStreamCaptureStatus st; hipStreamIsCapturing(s, &st); }
    printf("hipStreamIsCapturing:              %.2f us\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < 2000; ++i) { hipEventRecord(e, s); hipEventSynchronize(e); }
    printf("hipEventRecord+Synchronize (idle): %.2f us\n", (now() - t0) / 2000 * 1e6);
    t0 = now();
    for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, s, (int*)d); }
    double ti = now() - t0; hipStreamSynchronize(s);
    printf("launch issue:                      %.2f us (wall %.2f)\n", ti / 2000 * 1e6, (now() - t0) / 2000 * 1e6);
    t0 = now();
    for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, s, (int*)d); hipEventRecord(e, s); }
    ti = now() - t0; hipStreamSynchronize(s);
    printf("launch + event record issue:       %.2f us (wall %.2f)\n", ti / 2000 * 1e6, (now() - t0) / 2000 * 1e6);
    return 0;
}
