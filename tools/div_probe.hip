// Is rcp + 2 Newton steps + residual correction bit-identical to IEEE x/p on the value ranges of the path?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
__device__ __forceinline__ double fast_div(double x, double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0); r = __builtin_fma(r, e, r);
    double q = x * r;
    double rem = __builtin_fma(-p, q, x);
    return __builtin_fma(rem, r, q);
}
// one Newton step on the reciprocal instead of two; the quotient correction supplies the missing bits
__device__ __forceinline__ double fast_div1(double x, double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0); r = __builtin_fma(r, e, r);
    double q = x * r;
    double rem = __builtin_fma(-p, q, x);
    return __builtin_fma(rem, r, q);
}
// no Newton step on the reciprocal, two quotient corrections
__device__ __forceinline__ double fast_div2(double x, double p) {
    double r = __builtin_amdgcn_rcp(p);
    double q = x * r;
    double rem = __builtin_fma(-p, q, x);
    q = __builtin_fma(rem, r, q);
    rem = __builtin_fma(-p, q, x);
    return __builtin_fma(rem, r, q);
}
__global__ void k2(const double* x, const double* p, double* b1, double* b2, double* rr, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { b1[i] = fast_div1(x[i], p[i]); b2[i] = fast_div2(x[i], p[i]); rr[i] = __builtin_amdgcn_rcp(p[i]); }
}
__global__ void k(const double* x, const double* p, double* a, double* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = x[i] / p[i]; b[i] = fast_div(x[i], p[i]); }
}
int main() {
    int n = 1 << 24;
    std::vector<double> x(n), p(n), a(n), b(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0, 1);
    for (int i = 0; i < n; ++i) {
        int mode = i % 4;
        if (mode == 0) { x[i] = std::floor(U(rng) * 500); p[i] = U(rng) * 500 + 1e-3; }          // counts / reconstruction
        else if (mode == 1) { x[i] = 1.1920928955078125e-07; p[i] = std::exp(U(rng) * 40 - 25); }  // clipped zeros
        else if (mode == 2) { x[i] = std::exp(U(rng) * 60 - 30); p[i] = std::exp(U(rng) * 60 - 30); }
        else { x[i] = (i & 8) ? 0.0 : U(rng) * 1e6; p[i] = U(rng) * 1e-6 + 1e-12; }
    }
    double *dx, *dp, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dp, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long diff = 0, vs_host = 0; double maxrel = 0;
    for (int i = 0; i < n; ++i) {
        if (a[i] != b[i]) { ++diff; double r = std::fabs(a[i] - b[i]) / std::fabs(a[i]); if (r > maxrel) maxrel = r; }
        if (a[i] != x[i] / p[i]) ++vs_host;
    }
    {
        double *d1, *d2, *dr; hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dr, n * 8);
        k2<<<n / 256, 256>>>(dx, dp, d1, d2, dr, n);
        std::vector<double> b1(n), b2(n), rr(n);
        hipMemcpy(b1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rr.data(), dr, n * 8, hipMemcpyDeviceToHost);
        long m1 = 0, m2 = 0; double r1 = 0, r2 = 0, rcperr = 0;
        for (int i = 0; i < n; ++i) {
            if (a[i] != b1[i]) { ++m1; r1 = std::fmax(r1, std::fabs(a[i] - b1[i]) / std::fabs(a[i])); }
            if (a[i] != b2[i]) { ++m2; r2 = std::fmax(r2, std::fabs(a[i] - b2[i]) / std::fabs(a[i])); }
            rcperr = std::fmax(rcperr, std::fabs(rr[i] * p[i] - 1.0));
        }
        printf("v_rcp_f64 max |r p - 1| = %.3g\n", rcperr);
        printf("one Newton step + correction vs IEEE: %ld of %d differ (max rel %.3g)\n", m1, n, r1);
        printf("no Newton step, two corrections vs IEEE: %ld of %d differ (max rel %.3g)\n", m2, n, r2);
    }
    printf("fast_div vs device IEEE divide: %ld of %d differ (max rel %.3g); device IEEE vs host divide: %ld differ\n", diff, n, maxrel, vs_host);
    return 0;
}
