// Host driver of salamander_amd/csrc/salnmf_ncg_machine.h (the resumable Newton-CG the batched sample-embedding
// kernel runs): solves CorrNMF sample-embedding problems one at a time with a plain-loop evaluator, so that the CPU
// suite can check the state machine itself against scipy.optimize.minimize(method="Newton-CG") without a GPU.
// Test infrastructure only (compiled by tests/test_ncg_machine_host.py with g++; not part of libsalnmf.so).
//
// stdin (text): T dim N maxiter variance, then L[T][dim], so[T] (signature scalings), then per sample:
//   c[T] (sample scaling per term), a[T] (aux per term), x0[dim]
// stdout: per sample  status n_rounds x[dim]   (%.17g)
#include <cstdio>
#include <vector>

#include "../../salamander_amd/csrc/salnmf_ncg_machine.h"

struct HostVec {
    static constexpr int n = 64;
    double v[n];
    static double reduce(double s) { return s; }
};

int main() {
    int T, dim, N, maxiter;
    double variance;
    if (scanf("%d %d %d %d %lf", &T, &dim, &N, &maxiter, &variance) != 5 || dim > HostVec::n) return 2;
    std::vector<double> L(T * dim), so(T), c(T), a(T), hw(T), s(T), w(T);
    for (auto& x : L) if (scanf("%lf", &x) != 1) return 2;
    for (auto& x : so) if (scanf("%lf", &x) != 1) return 2;
    if (maxiter <= 0) maxiter = 200 * dim;
    using namespace salnmf::ncgm;
    for (int n = 0; n < N; ++n) {
        for (auto& x : c) if (scanf("%lf", &x) != 1) return 2;
        for (auto& x : a) if (scanf("%lf", &x) != 1) return 2;
        HostVec x0{}, sg{};
        for (int m = 0; m < dim; ++m) if (scanf("%lf", &x0.v[m]) != 1) return 2;
        for (int i = 0; i < T; ++i)
            for (int m = 0; m < dim; ++m) sg.v[m] += a[i] * L[i * dim + m];
        Machine<HostVec> mc;
        mc.begin(x0);
        int rounds = 0;
        while (!mc.finished()) {
            const int req = mc.request();
            const HostVec& y = req == REQ_HESSP ? mc.ps : mc.yv;
            for (int i = 0; i < T; ++i) {
                double acc = 0.0;
                for (int m = 0; m < dim; ++m) acc += L[i * dim + m] * y.v[m];
                s[i] = acc;
            }
            HostVec r{};
            double f = 0.0;
            if (req == REQ_POINT) {
                double lin = 0.0, ex = 0.0, yy = 0.0;
                for (int i = 0; i < T; ++i) {
                    w[i] = hw[i] = exp((c[i] + so[i]) + s[i]);
                    lin += s[i] * a[i];
                    ex += w[i];
                }
                for (int m = 0; m < dim; ++m) yy += y.v[m] * y.v[m];
                double v = lin;
                v -= ex;
                v -= yy / (2 * variance);
                f = -v;
                for (int m = 0; m < dim; ++m) {
                    double comb = 0.0;
                    for (int i = 0; i < T; ++i) comb += w[i] * L[i * dim + m];
                    double g = -comb;
                    g += sg.v[m];
                    g -= y.v[m] / variance;
                    r.v[m] = -g;
                }
            } else {
                for (int m = 0; m < dim; ++m) {
                    double comb = 0.0;
                    for (int i = 0; i < T; ++i) comb += (hw[i] * s[i]) * L[i * dim + m];
                    r.v[m] = comb + y.v[m] / variance;
                }
            }
            mc.advance(f, r, dim, maxiter);
            ++rounds;
        }
        printf("%d %d", mc.status, rounds);
        for (int m = 0; m < dim; ++m) printf(" %.17g", mc.xk.v[m]);
        printf("\n");
    }
    return 0;
}
