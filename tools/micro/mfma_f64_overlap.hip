// Does the fp64 matrix instruction (v_mfma_f64_16x16x4_f64, 2 048 flop per wave instruction) run BESIDE the fp64 vector pipe on gfx950,
// or on it?  MI355X quotes the same 78.6 TFLOP/s for both, so fp64 work moved from v_fma_f64 to MFMA gains only if the two overlap.
//   mode 0: every wave a stream of vector FMAs (8 independent chains)                     -> T_v
//   mode 1: every wave a stream of MFMAs (4 independent accumulator tiles)                -> T_m
//   mode 2: waves 0-3 of a 512-thread workgroup vector, waves 4-7 MFMA (one of each per SIMD), full work each
//   mode 3: every wave both, interleaved in one loop body (independent streams), full work each
// Iteration counts are set so that the same flop count goes through either pipe.  If modes 2 / 3 take about max(T_v, T_m) the pipes
// run side by side; if about T_v + T_m they are one pipe.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_overlap.hip -o mfma_f64_overlap && ./mfma_f64_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_mix(double* out, int mode, int iters, double a, double b) {
    const int wave = threadIdx.x >> 6;
    // (512 threads: waves w and w + 4 share a SIMD; 256 threads: one wave per SIMD per workgroup, roles swapped in odd workgroups)
    const bool vec_role = THREADS == 512 ? wave < 4 : ((wave < 2) != bool(blockIdx.x & 1));
    const bool do_v = mode == 0 || mode == 3 || (mode == 2 && vec_role);
    const bool do_m = mode == 1 || mode == 3 || (mode == 2 && !vec_role);
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    const double ma = 1.0 + threadIdx.x * 1e-6, mb = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
        if (do_v) {
            // 64 wave-level FMAs = 64 x 128 flop = 8 192 flop per wave
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
            }
        }
        if (do_m) {
            // 4 MFMAs = 4 x 2 048 flop = 8 192 flop per wave
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS>
static void run(const char* label, int grid) {
    double* out;
    hipMalloc(&out, sizeof(double) * grid * THREADS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float ms[4];
    for (int mode = 0; mode < 4; ++mode) {
        k_mix<THREADS><<<grid, THREADS>>>(out, mode, 100, 0.999999, 1e-9);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_mix<THREADS><<<grid, THREADS>>>(out, mode, iters, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double waves = double(grid) * THREADS / 64;
    const double tf_v = waves * iters * 8192.0 / (ms[0] * 1e-3) / 1e12, tf_m = waves * iters * 8192.0 / (ms[1] * 1e-3) / 1e12;
    // mode 2: half the waves each role; mode 3: every wave both roles
    printf("%s: vector only %.3f ms (%.1f TFLOP/s)  MFMA only %.3f ms (%.1f TFLOP/s)\n", label, ms[0], tf_v, ms[1], tf_m);
    printf("    half the waves vector + half MFMA: %.3f ms  (side by side would be %.3f, one pipe %.3f)\n", ms[2],
           0.5 * (ms[0] > ms[1] ? ms[0] : ms[1]), 0.5 * (ms[0] + ms[1]));
    printf("    every wave both, interleaved:      %.3f ms  (side by side would be %.3f, one pipe %.3f)\n", ms[3],
           ms[0] > ms[1] ? ms[0] : ms[1], ms[0] + ms[1]);
    hipFree(out);
}
int main() {
    run<512>("512-thread workgroups, 2 per CU (4 waves per SIMD)", 512);
    run<512>("512-thread workgroups, 1 per CU (2 waves per SIMD)", 256);
    run<256>("256-thread workgroups, 8 per CU (8 waves per SIMD)", 2048);
    return 0;
}
