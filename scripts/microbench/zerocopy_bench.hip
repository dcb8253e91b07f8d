// Host round trip of one small device call, by how its payload travels -- diagnostic.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/microbench/zerocopy_bench.bin scripts/microbench/zerocopy_bench.hip && scripts/microbench/zerocopy_bench.bin
//   A  hipMemcpyAsync H2D + kernel + hipMemcpyAsync D2H + hipStreamSynchronize          (what the library's entry points do)
//   B  kernel reads its input from mapped pinned host memory and writes its output there + hipStreamSynchronize
//   C  H2D copy + kernel writing its output to mapped host memory + hipStreamSynchronize
//   D  like A with four D2H copies                                                        (the PARSAC batch: results, models, masks, bins)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void work(const double *in, double *out, int n, int spin) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += in[i];
    for (int k = 0; k < spin; ++k) acc = acc * 1.0000001 + 1e-9;   // stands for the solve
    for (int i = threadIdx.x; i < n; i += 256) out[i] = acc + in[i];
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int bytes : {4096, 40960, 262144}) {
        const int n = bytes / 8;
        double *d_in, *d_out, *h_in, *h_out, *h_in_dev, *h_out_dev;
        CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_out, bytes));
        CK(hipHostMalloc(&h_in, bytes, hipHostMallocMapped)); CK(hipHostMalloc(&h_out, bytes, hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void **)&h_in_dev, h_in, 0)); CK(hipHostGetDevicePointer((void **)&h_out_dev, h_out, 0));
        for (int i = 0; i < n; ++i) h_in[i] = 1.0;
        const char *names[4] = {"A copies both ways", "B zero-copy both ways", "C copy in, zero-copy out", "D copy in, four copies out"};
        for (int mode = 0; mode < 4; ++mode) {
            const int N = 1000;
            double tot = 0;
            for (int it = 0; it < N + 50; ++it) {
                const auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) {
                    CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d_in, d_out, n, 2000);
                    CK(hipMemcpyAsync(h_out, d_out, bytes / 8, hipMemcpyDeviceToHost, st));
                } else if (mode == 1) {
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, h_in_dev, h_out_dev, n, 2000);
                } else if (mode == 2) {
                    CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d_in, h_out_dev, n, 2000);
                } else {
                    CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, st, d_in, d_out, n, 2000);
                    for (int q = 0; q < 4; ++q) CK(hipMemcpyAsync(h_out + q * (n / 32), d_out + q * (n / 32), bytes / 32, hipMemcpyDeviceToHost, st));
                }
                CK(hipStreamSynchronize(st));
                if (it >= 50) tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            std::printf("%7d B in: %-28s %7.2f us per call\n", bytes, names[mode], 1e6 * tot / N);
        }
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipHostFree(h_in); (void)hipHostFree(h_out);
    }
    return 0;
}
