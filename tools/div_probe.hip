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
    printf("fast_div vs device IEEE divide: %ld of %d differ (max rel %.3g); device IEEE vs host divide: %ld differ\n", diff, n, maxrel, vs_host);
    return 0;
}
