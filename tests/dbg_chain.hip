#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(const double* o_, const double* v_, double* out_asm, double* out_ref, double start) {
    const int lane = threadIdx.x;
    double mine = start, sum = start, ref = 0.0;
    for (int j0 = 0; j0 < 64; j0 += 8) {
        double o[8], v[8];
        for (int u = 0; u < 8; ++u) { o[u] = o_[j0 + u]; v[u] = v_[j0 + u]; }
        for (int u = 0; u < 8; ++u) { ref = lane == j0 + u ? sum : ref; sum = sum - o[u]; sum = sum + v[u]; }
        const unsigned long long live = ~0ull << j0;
        unsigned long long saved;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b64 exec, %[lv]\n\t"
#define STEP(U) "s_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[m], %[m], -%[o" #U "]\n\t" "v_add_f64 %[m], %[m], %[v" #U "]\n\t"
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
            "s_mov_b64 exec, %[sv]"
            : [m] "+v"(mine), [sv] "=&s"(saved)
            : [lv] "s"(live), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [o4] "v"(o[4]),
              [o5] "v"(o[5]), [o6] "v"(o[6]), [o7] "v"(o[7]), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]),
              [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
            : "scc");
    }
    out_asm[lane] = mine; out_ref[lane] = ref;
}
int main() {
    double ho[64], hv[64], ha[64], hr[64];
    for (int i = 0; i < 64; ++i) { ho[i] = 999.0; hv[i] = (rand() % 1000) / 37.0; }
    double *o, *v, *a, *r;
    hipMalloc(&o, 512); hipMalloc(&v, 512); hipMalloc(&a, 512); hipMalloc(&r, 512);
    hipMemcpy(o, ho, 512, hipMemcpyHostToDevice); hipMemcpy(v, hv, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(o, v, a, r, 999.0 * 160);
    hipMemcpy(ha, a, 512, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (ha[i] != hr[i]) { if (bad < 8) printf("lane %d asm %.6f ref %.6f\n", i, ha[i], hr[i]); ++bad; }
    printf("mismatches %d\n", bad);
    return 0;
}
