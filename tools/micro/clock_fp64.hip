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
// Do the two pipes overlap?  (round 5)  One kernel, three roles per wave: FMA only (the loop of k_fma), LDS only (the loop of k_lds),
// or BOTH in the same loop body (independent instruction streams inside one wave).  mode 0: every wave FMA; 1: every wave LDS;
// 2: waves 0-1 of a workgroup FMA, waves 2-3 LDS (roles swapped in odd workgroups so that every SIMD sees both); 3: every wave both,
// half the iterations each.  The iteration counts are chosen so that modes 0 and 1 take the same time T: if the pipes run side by
// side, modes 2 and 3 (half the FMA work + half the LDS work) take about T/2 (the LDS role is a pure stream of reads: inline asm, nothing consumes them); if one waits for the other, T.  `lds_bytes` of dynamic LDS
// sets the occupancy (16 KB: 8 workgroups per CU; 53 KB: 3, the occupancy of k_fine_cert / k_post_chain_r).
// W: bytes per lane of the LDS read (16: ds_read_b128, 8: b64, 4: b32); F32: the FMA role in packed fp32 instead of fp64
template <int W, bool F32>
__global__ void __launch_bounds__(256) k_mix(double* out, int mode, int it_f, int it_l, double a, double b) {
    extern __shared__ double2 shm[];
    for (int i = threadIdx.x; i < 1024; i += 256) shm[i] = make_double2(i, -i);
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const bool odd = blockIdx.x & 1;
    const bool do_f = mode == 0 || mode == 3 || (mode == 2 && ((wave < 2) != odd));
    const bool do_l = mode == 1 || mode == 3 || (mode == 2 && ((wave < 2) == odd));
    const int nf = do_f ? (mode == 3 ? it_f / 2 : it_f) : 0, nl = do_l ? (mode == 3 ? it_l / 2 : it_l) : 0;
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double2 acc = make_double2(0, 0);
    float facc = 0.f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f xf[8];
    for (int i = 0; i < 8; ++i) xf[i] = v2f{(float)threadIdx.x * 1e-3f + i, 1.0f + i};
    const v2f fa = {(float)a, (float)a}, fb = {(float)b, (float)b};
    int idx = threadIdx.x;
    const unsigned lds_base = (unsigned)(uintptr_t)shm;          // (an LDS pointer's low 32 bits are its LDS byte address)
    const int n = nf > nl ? nf : nl;
    for (int it = 0; it < n; ++it) {
        if (it < nl) {                                        // (wave-uniform)
            // a pure LDS stream: 16 reads in flight, no vector instruction consumes them (inline asm: the compiler cannot drop them)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const unsigned ad = lds_base + ((unsigned)(idx * W) & 1023u);
#define RD(u) do { if (W == 16) { v4u r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"((u) * 1024)); } \
                   else if (W == 8) { v2u r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"((u) * 1024)); } \
                   else { unsigned r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"((u) * 1024)); } } while (0)
            RD(0); RD(1); RD(2); RD(3); RD(4); RD(5); RD(6); RD(7); RD(8); RD(9); RD(10); RD(11); RD(12); RD(13); RD(14); RD(15);
#undef RD
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            idx = (idx + 5) & 1023;
        }
        if (it < nf) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (F32) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xf[i] = xf[i] * fa + fb;      // (v_pk_fma_f32: two lanes' worth per instruction)
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
                }
            }
        }
    }
    double s = acc.x + acc.y + facc; for (int i = 0; i < 8; ++i) s += x[i] + xf[i].x + xf[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int W, bool F32>
static void mix_table(double* out, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds_sizes[2] = {16 * 1024, 53 * 1024};
    for (int oc = 0; oc < 2; ++oc) {
        (void)hipFuncSetAttribute((const void*)k_mix<W, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        // calibrate: LDS iterations that take as long as it_f FMA iterations (each FMA iteration here: 32 wave-FMAs; each LDS: 16 b128 reads + 32 adds)
        const int it_f = 40000;
        auto run = [&](int mode, int itf, int itl) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL((k_mix<W, F32>), dim3(blocks), dim3(256), lds_sizes[oc], 0, out, mode, itf, itl, 1.0000001, 1e-9);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            return ms;
        };
        const float tf = run(0, it_f, 0);
        int it_l = 10000;
        float tl = run(1, 0, it_l);
        it_l = (int)(it_l * (tf / tl));
        tl = run(1, 0, it_l);
        const float t2 = run(2, it_f, it_l), t3 = run(3, it_f, it_l);
        printf("pipes [%s FMA | ds_read_b%d], %2zu KB LDS per workgroup (%s): all-FMA %.2f ms | all-LDS %.2f ms | half the waves each %.2f ms | every wave both, half each %.2f ms   (side by side: %.2f; one after the other: %.2f)\n",
               F32 ? "packed fp32" : "fp64", 8 * W, lds_sizes[oc] / 1024, oc ? "3 workgroups per CU" : "8 per CU", tf, tl, t2, t3, 0.5f * (tf > tl ? tf : tl), 0.5f * (tf + tl));
    }
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
    mix_table<16, false>(out, blocks);
    mix_table<8, false>(out, blocks);
    mix_table<4, false>(out, blocks);
    mix_table<16, true>(out, blocks);
    mix_table<4, true>(out, blocks);
    return 0;
}
