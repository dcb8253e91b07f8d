// Host round-trip cost of "H2D + tiny kernel + D2H + wait" with different waits -- diagnostic.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <atomic>
#include <vector>
__global__ void tiny(double *p, int n) { if (threadIdx.x < n) p[threadIdx.x] += 1.0; }
__global__ void tiny_flag(double *p, int n, volatile unsigned *flag, unsigned seq) {
    if (threadIdx.x < n) p[threadIdx.x] += 1.0;
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *d, *hin, *hout; unsigned *flag_h; unsigned *flag_d;
    CK(hipMalloc(&d, 4096)); CK(hipHostMalloc(&hin, 4096)); CK(hipHostMalloc(&hout, 4096));
    CK(hipHostMalloc(&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&flag_d, flag_h, 0));
    *flag_h = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        double tot = 0;
        for (int i = 0; i < N + 100; ++i) {
            auto t0 = now();
            CK(hipMemcpyAsync(d, hin, 2048, hipMemcpyHostToDevice, st));
            if (mode == 0) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, 64);
                CK(hipMemcpyAsync(hout, d, 2048, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            } else if (mode == 1) {   // event + query spin
                static hipEvent_t ev = nullptr; if (!ev) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, 64);
                CK(hipMemcpyAsync(hout, d, 2048, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev, st));
                while (hipEventQuery(ev) == hipErrorNotReady) {}
            } else if (mode == 2) {   // stream write value + spin on mapped host word
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, 64);
                CK(hipMemcpyAsync(hout, d, 2048, hipMemcpyDeviceToHost, st));
                hipError_t e = hipStreamWriteValue32(st, flag_d, (unsigned)(i + 1), 0);
                if (e != hipSuccess) { std::printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); break; }
                while (*(volatile unsigned *)flag_h != (unsigned)(i + 1)) {}
            } else {                  // kernel writes results + flag straight to mapped host memory
                static double *hmap = nullptr, *dmap = nullptr;
                if (!hmap) { CK(hipHostMalloc(&hmap, 4096, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void **)&dmap, hmap, 0)); }
                hipLaunchKernelGGL(tiny_flag, dim3(1), dim3(64), 0, st, dmap, 64, flag_d, (unsigned)(1000000 + i));
                while (*(volatile unsigned *)flag_h != (unsigned)(1000000 + i)) {}
            }
            if (i >= 100) tot += std::chrono::duration<double>(now() - t0).count();
        }
        const char *names[4] = {"hipStreamSynchronize", "event query spin", "stream write value + spin", "kernel writes mapped host + flag"};
        std::printf("%-36s %.2f us per round trip\n", names[mode], 1e6 * tot / N);
    }
    // the same with a second thread hammering another stream (contention on the runtime)
    std::atomic<bool> stop{false};
    std::thread other([&] {
        hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        double *d2; (void)hipMalloc(&d2, 4096);
        while (!stop.load()) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d2, 64); (void)hipStreamSynchronize(s2); }
    });
    for (int mode = 0; mode < 1; ++mode) {
        double tot = 0;
        for (int i = 0; i < N + 100; ++i) {
            auto t0 = now();
            CK(hipMemcpyAsync(d, hin, 2048, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, 64);
            CK(hipMemcpyAsync(hout, d, 2048, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (i >= 100) tot += std::chrono::duration<double>(now() - t0).count();
        }
        std::printf("%-36s %.2f us per round trip (second thread busy on another stream)\n", "hipStreamSynchronize", 1e6 * tot / N);
    }
    stop.store(true); other.join();
    return 0;
}
