// Accuracy of the fused log(x/p) used by the objective kernels vs a long-double host reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include "../salamander_amd/csrc/salnmf_kernels.h"
__global__ void k(const double* x, const double* p, double* a, double* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = (salnmf::log_operand_ok(x[i]) && salnmf::log_operand_ok(p[i])) ? salnmf::log_ratio(x[i], p[i]) : log(x[i] / p[i]); b[i] = log(x[i] / p[i]); }
}
// log_pos (table driven, the objectives' logarithm) against the library log and a long-double reference
__global__ void kpos(const double* p, double* a, double* b, int n) {
    __shared__ double tab[salnmf::LOGTAB_DOUBLES];
    salnmf::stage_logtab(tab, threadIdx.x);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = salnmf::log_pos_ok(p[i]) ? salnmf::log_pos(p[i], tab) : log(p[i]); b[i] = log(p[i]); }
}
int main() {
    int n = 1 << 22;
    std::vector<double> x(n), p(n), a(n), b(n);
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0, 1);
    for (int i = 0; i < n; ++i) {
        int mode = i % 5;
        if (mode == 0) { x[i] = std::floor(U(rng) * 500) + 1; p[i] = U(rng) * 500 + 1e-3; }
        else if (mode == 1) { x[i] = 1.1920928955078125e-07; p[i] = std::exp(U(rng) * 40 - 25); }
        else if (mode == 2) { x[i] = std::exp(U(rng) * 200 - 100); p[i] = std::exp(U(rng) * 200 - 100); }
        else if (mode == 3) { x[i] = U(rng) * 100 + 1; p[i] = x[i] * (1 + (U(rng) - 0.5) * 1e-6); }   // ratio ~ 1
        else { x[i] = U(rng) * 1e4; p[i] = x[i] * std::exp((U(rng) - 0.5) * 1.5); }                    // around the sqrt(2) seams
    }
    double *dx, *dp, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dp, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    double mrel_a[5] = {0}, mrel_b[5] = {0}, mabs_a[5] = {0}; int worst[5] = {0};
    for (int i = 0; i < n; ++i) {
        long double ref = logl((long double)x[i] / (long double)p[i]);
        if (ref == 0) continue;
        int m = i % 5;
        double ea = std::fabs((double)(((long double)a[i] - ref) / ref)), eb = std::fabs((double)(((long double)b[i] - ref) / ref));
        if (ea > mrel_a[m]) { mrel_a[m] = ea; worst[m] = i; }
        if (eb > mrel_b[m]) mrel_b[m] = eb;
        double aa = std::fabs((double)((long double)a[i] - ref)); if (aa > mabs_a[m]) mabs_a[m] = aa;
    }
    for (int m = 0; m < 5; ++m) { int i = worst[m]; printf("mode %d: log_ratio max rel err %.3g (abs %.3g) | ocml log(x/p) max rel err %.3g | worst x=%.17g p=%.17g got %.17g ref %.17Lg\n", m, mrel_a[m], mabs_a[m], mrel_b[m], x[i], p[i], a[i], logl((long double)x[i] / (long double)p[i])); }
    // ---- log_pos
    for (int i = 0; i < n; ++i) {
        int mode = i % 5;
        if (mode == 0) p[i] = U(rng) * 500 + 1e-3;
        else if (mode == 1) p[i] = std::exp(U(rng) * 1400 - 700);
        else if (mode == 2) p[i] = 1 + (U(rng) - 0.5) * 1e-6;
        else if (mode == 3) p[i] = std::ldexp(1 + std::floor(U(rng) * 257) / 256.0, (int)(U(rng) * 40) - 20) * (1 + (U(rng) - 0.5) * 1e-12);  // table seams
        else p[i] = 1.1920928955078125e-07 * (1 + U(rng));
    }
    hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice);
    kpos<<<n / 256, 256>>>(dp, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    double ea[5] = {0}, eb[5] = {0};
    for (int i = 0; i < n; ++i) {
        long double ref = logl((long double)p[i]);
        long double sc = std::max(fabsl(ref), (long double)0.5);
        int m = i % 5;
        ea[m] = std::max(ea[m], (double)(fabsl((long double)a[i] - ref) / sc));
        eb[m] = std::max(eb[m], (double)(fabsl((long double)b[i] - ref) / sc));
    }
    for (int m = 0; m < 5; ++m) printf("log_pos mode %d: max |err| / max(|log p|, 0.5) = %.3g | ocml log %.3g\n", m, ea[m], eb[m]);
    return 0;
}
