// Shader clock under load: a kernel that saturates the fp64 vector pipes (8 independent FMA chains per lane) reads the shader
// cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its loop: cycles / real time = the clock the
// SIMDs actually ran at, and lane-FMAs / real time = achieved TFLOP/s.  Same for an LDS-read-bound loop.
//   hipcc --offload-arch=gfx950 -O3 clock_fp64.hip -o clock_fp64 && ./clock_fp64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(256) k_fma(double* out, unsigned long long* stamps, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}
__global__ void __launch_bounds__(256) k_lds(double* out, unsigned long long* stamps, int iters) {
    __shared__ double2 sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = make_double2(i, -i);
    __syncthreads();
    double2 acc = make_double2(0, 0);
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const double2 v = sh[(idx + 256 * u) & 4095]; acc.x += v.x; acc.y += v.y; }
        idx = (idx + 5) & 4095;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int blocks = 256 * 8;
    double* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * sizeof(double)); hipMalloc(&st, blocks * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(blocks * 2);
    for (int pass = 0; pass < 3; ++pass) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        const double mhz = cyc / wall * 100.0;
        const double flop = 2.0 * blocks * 256.0 * iters * 128.0;
        printf("fp64 FMA: %.2f ms  %.1f TFLOP/s  shader clock %.0f MHz  (cycles per wave-FMA at 8 waves/SIMD... %.2f)\n", ms, flop / ms / 1e9, mhz,
               (cyc / blocks) / (iters * 128.0) / 2.0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, out, st, iters / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        cyc = 0; wall = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        const double bytes = 16.0 * blocks * 256.0 * (iters / 4) * 16.0;
        printf("LDS b128: %.2f ms  %.1f TB/s  shader clock %.0f MHz  => %.1f B/clk/CU\n", ms, bytes / ms / 1e9, cyc / wall * 100.0,
               bytes / (ms * 1e-3) / 256.0 / (cyc / wall * 1e8));
    }
    return 0;
}
