// The Newton-CG of salnmf_newtoncg.h as a RESUMABLE state machine: one evaluation request at a time.
//
// salnmf_newtoncg.h runs a solve as nested loops around an evaluator and therefore needs a wavefront per problem.
// The batched sample-embedding kernel (salnmf_corr_batched.h) advances SIXTEEN problems per wavefront in lockstep
// rounds -- one evaluation per problem and round, all sixteen evaluated together on the fp64 MFMA units -- so the
// solver must be able to stop at every evaluation and resume with its result.  This header is that form: the same
// algorithm (SciPy's `_minimize_newtoncg`, the More-Thuente search `dcsrch`/`dcstep` it tries first and the
// bracketing / zoom search it falls back to; constants, order of tests and status codes as in salnmf_newtoncg.h,
// whose helper functions it reuses unchanged), cut at its evaluation calls.
//
// Differences in execution, none in the iterates' definition:
//   * every point evaluation returns objective AND gradient (on the device both come out of the same two matrix
//     products), so where the loop form calls fun(y) and later grad(y) at the same point (the bracketing and zoom
//     searches) one request serves both;
//   * `prepare_hess(xk)` needs no request: the Hessian weights exp(.) at xk are those of the last point evaluation,
//     which -- in every path through the line searches -- is the accepted point (the evaluator keeps them).
//
// The vector type V holds the components one execution lane owns (on the device 4 * DT components of one of the 16
// problems, with the 4 lanes of a problem reducing through V::reduce; on the host -- tests/native/ncg_machine_host.cpp,
// which runs this very header against SciPy in the CPU suite -- all components, reduce = identity).
#pragma once
#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SALNMF_HD __host__ __device__ __forceinline__
#else
#define SALNMF_HD inline
#endif

namespace salnmf {
namespace ncgm {

enum Status { OK = 0, MAXITER = 1, LINESEARCH_FAILED = 2, CG_FAILED = 3 };  // = ncg::Status

enum Request { REQ_NONE = 0, REQ_POINT = 1, REQ_HESSP = 2 };

// phases: < FIRST_TRANSIENT wait for an evaluation (or are terminal); the others run without one
enum Phase {
    EMPTY = 0,      // no problem in this slot
    DONE,           // solved: result in xk, status set
    WAIT_PREP,      // (device only) the slot's problem is being prepared by a round of its own
    WAIT_INIT,      // f, g at the start point
    WAIT_CG_HP,     // Hessian . psupi
    WAIT_W1,        // More-Thuente trial point
    WAIT_W2_FIRST,  // bracketing search: first trial point
    WAIT_W2,        // bracketing search: doubled trial point
    WAIT_ZOOM,      // zoom trial point
    FIRST_TRANSIENT,
    ACCEPT = FIRST_TRANSIENT,
    NEWTON_TOP,
    CG_START,
    CG_TOP,
    CG_END,
    W1_START,
    W1_TOP,
    W2_START,
    W2_TOP,
    ZOOM_TOP,
};

SALNMF_HD double m_sign(double v) { return (double)((v > 0.0) - (v < 0.0)); }
SALNMF_HD double m_clamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
// x + fl(t p): the product rounded first, so that the accepted iterate has the bits of the accepted trial point
SALNMF_HD double m_point(double x, double t, double p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(x, __dmul_rn(t, p));
#else
    volatile double tp = t * p;  // (volatile: no contraction into a fused multiply-add on the host either)
    return x + tp;
#endif
}
SALNMF_HD double m_mul(double t, double p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dmul_rn(t, p);
#else
    volatile double tp = t * p;
    return tp;
#endif
}

// MINPACK-2 dcstep (the same code as ncg::dcstep, on the machine's scalar block)
struct Step {
    double stx, fx, dx, sty, fy, dy, stp;
    bool brackt;
};
SALNMF_HD void m_dcstep(Step& s, double fp, double dp, double stpmin, double stpmax) {
    const double sgnd = m_sign(dp) * m_sign(s.dx);
    double stpf;
    const double stp = s.stp;
    if (fp > s.fx) {
        const double theta = 3.0 * (s.fx - fp) / (stp - s.stx) + s.dx + dp;
        const double sc = fmax(fmax(fabs(theta), fabs(s.dx)), fabs(dp));
        double gamma = sc * sqrt((theta / sc) * (theta / sc) - (s.dx / sc) * (dp / sc));
        if (stp < s.stx) gamma = -gamma;
        const double p = (gamma - s.dx) + theta;
        const double q = ((gamma - s.dx) + gamma) + dp;
        const double r = p / q;
        const double stpc = s.stx + r * (stp - s.stx);
        const double stpq = s.stx + ((s.dx / ((s.fx - fp) / (stp - s.stx) + s.dx)) / 2.0) * (stp - s.stx);
        stpf = (fabs(stpc - s.stx) <= fabs(stpq - s.stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
        s.brackt = true;
    } else if (sgnd < 0.0) {
        const double theta = 3.0 * (s.fx - fp) / (stp - s.stx) + s.dx + dp;
        const double sc = fmax(fmax(fabs(theta), fabs(s.dx)), fabs(dp));
        double gamma = sc * sqrt((theta / sc) * (theta / sc) - (s.dx / sc) * (dp / sc));
        if (stp > s.stx) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = ((gamma - dp) + gamma) + s.dx;
        const double r = p / q;
        const double stpc = stp + r * (s.stx - stp);
        const double stpq = stp + (dp / (dp - s.dx)) * (s.stx - stp);
        stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
        s.brackt = true;
    } else if (fabs(dp) < fabs(s.dx)) {
        const double theta = 3.0 * (s.fx - fp) / (stp - s.stx) + s.dx + dp;
        const double sc = fmax(fmax(fabs(theta), fabs(s.dx)), fabs(dp));
        double gamma = sc * sqrt(fmax(0.0, (theta / sc) * (theta / sc) - (s.dx / sc) * (dp / sc)));
        if (stp > s.stx) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = (gamma + (s.dx - dp)) + gamma;
        const double r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (s.stx - stp);
        else if (stp > s.stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - s.dx)) * (s.stx - stp);
        if (s.brackt) {
            stpf = (fabs(stpc - stp) < fabs(stpq - stp)) ? stpc : stpq;
            if (stp > s.stx) stpf = fmin(stp + 0.66 * (s.sty - stp), stpf);
            else stpf = fmax(stp + 0.66 * (s.sty - stp), stpf);
        } else {
            stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
            stpf = m_clamp(stpf, stpmin, stpmax);
        }
    } else {
        if (s.brackt) {
            const double theta = 3.0 * (fp - s.fy) / (s.sty - stp) + s.dy + dp;
            const double sc = fmax(fmax(fabs(theta), fabs(s.dy)), fabs(dp));
            double gamma = sc * sqrt((theta / sc) * (theta / sc) - (s.dy / sc) * (dp / sc));
            if (stp > s.sty) gamma = -gamma;
            const double p = (gamma - dp) + theta;
            const double q = ((gamma - dp) + gamma) + s.dy;
            const double r = p / q;
            stpf = stp + r * (s.sty - stp);
        } else if (stp > s.stx) {
            stpf = stpmax;
        } else {
            stpf = stpmin;
        }
    }
    if (fp > s.fx) {
        s.sty = stp; s.fy = fp; s.dy = dp;
    } else {
        if (sgnd < 0.0) { s.sty = s.stx; s.fy = s.fx; s.dy = s.dx; }
        s.stx = stp; s.fx = fp; s.dx = dp;
    }
    s.stp = stpf;
}

SALNMF_HD bool m_cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double& xmin) {
    const double C = fpa, db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    const double r0 = fb - fa - C * db, r1 = fc - fa - C * dc;
    double A = dc * dc * r0 + (-(db * db)) * r1;
    double B = (-(dc * dc * dc)) * r0 + (db * db * db) * r1;
    if (denom == 0.0) return false;
    A /= denom;
    B /= denom;
    const double radical = B * B - 3 * A * C;
    if (!(radical >= 0.0) || 3 * A == 0.0) return false;
    xmin = a + (-B + sqrt(radical)) / (3 * A);
    return isfinite(xmin);
}
SALNMF_HD bool m_quadmin(double a, double fa, double fpa, double b, double fb, double& xmin) {
    const double D = fa, C = fpa, db = b - a;
    if (db * db == 0.0) return false;
    const double B = (fb - D - C * db) / (db * db);
    if (2.0 * B == 0.0) return false;
    xmin = a - C / (2.0 * B);
    return isfinite(xmin);
}

// V: struct { static constexpr int n; double v[n]; static double reduce(double partial); }
template <class V>
SALNMF_HD double v_dot(const V& a, const V& b) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < V::n; ++i) s = fma(a.v[i], b.v[i], s);
    return V::reduce(s);
}
template <class V>
SALNMF_HD double v_l1(const V& a) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < V::n; ++i) s += fabs(a.v[i]);
    return V::reduce(s);
}

template <class V>
struct Machine {
    // vectors: iterate; gradient at xk (then the accepted point's); CG solution = search direction; CG residual;
    // CG direction = the Hessian-vector request; the requested point of a line search
    V xk, gv, xs, ri, ps, yv;
    double old_fval, old_old_fval, update_l1, dri0, termcond, derphi0, new_fval, alphak;
    double ri_l1;  // |ri|_1, formed next to <ri, ri> (two independent reductions side by side) for the test at CG_TOP
    // line-search scalars.  More-Thuente: stx fx dx sty fy dy | stmin stmax width width1 stp gtest.
    // Bracketing: alpha0 alpha1 phi_a1 phi_a0 derphi_a0; zoom: a_lo a_hi phi_lo phi_hi derphi_lo phi_rec a_rec a_j.
    double u[13];
    int phase, k, cg_i, cg_k2, it, status;
    bool have_old_old, brackt, stage2;

    enum { STX = 0, FX, DX, STY, FY, DY, STMIN, STMAX, WIDTH, WIDTH1, STP, GTEST };
    enum { ALPHA0 = 0, ALPHA1, PHI_A1, PHI_A0, DERPHI_A0, A_LO, A_HI, PHI_LO, PHI_HI, DERPHI_LO, PHI_REC, A_REC, A_J };

    SALNMF_HD int request() const {
        return phase == WAIT_CG_HP ? REQ_HESSP : (phase >= WAIT_INIT && phase < FIRST_TRANSIENT) ? REQ_POINT : REQ_NONE;
    }
    SALNMF_HD bool finished() const { return phase == DONE; }

    // start a solve at x0: the first request is the point x0 itself
    SALNMF_HD void begin(const V& x0) {
        xk = x0;
        yv = x0;
        phase = WAIT_INIT;
        k = 0;
        status = OK;
        have_old_old = false;
        old_old_fval = 0.0;
        update_l1 = DBL_MAX;
    }

    SALNMF_HD void zoom_init(double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo) {
        u[A_LO] = a_lo; u[A_HI] = a_hi; u[PHI_LO] = phi_lo; u[PHI_HI] = phi_hi; u[DERPHI_LO] = derphi_lo;
        u[PHI_REC] = old_fval;
        u[A_REC] = 0.0;
        it = 0;
        phase = ZOOM_TOP;
    }
    SALNMF_HD void accept(double step, double fval, const V& g) {
        alphak = step;
        new_fval = fval;
        gv = g;
        phase = ACCEPT;
    }
    SALNMF_HD void set_point(double t) {
#pragma unroll
        for (int j = 0; j < V::n; ++j) yv.v[j] = m_point(xk.v[j], t, xs.v[j]);
    }

    // Deliver the result of the pending request (f: objective, for REQ_POINT; r: gradient or Hessian-vector product)
    // and run until the next request or the end.  dim = length of the problem's vectors, maxiter = Newton iterations.
    SALNMF_HD void advance(double f, const V& r, int dim, int maxiter) {
        constexpr double c1 = 1e-4, c2 = 0.9;
        const double phi0 = old_fval;  // (of the current Newton iteration; the WAIT_INIT handler sets old_fval itself)
        if (phase == WAIT_INIT) {
            old_fval = f;
            gv = r;
            phase = NEWTON_TOP;
        } else if (phase == WAIT_CG_HP) {
            const double curv = v_dot(ps, r);
            if (0 <= curv && curv <= 3 * DBL_EPSILON) {
                phase = CG_END;
            } else if (curv < 0) {
                if (cg_i == 0) {
                    const double sdesc = dri0 / (-curv);  // steepest descent fallback: xsupi = dri0 / (-curv) * b
#pragma unroll
                    for (int j = 0; j < V::n; ++j) xs.v[j] = sdesc * -gv.v[j];
                }
                phase = CG_END;
            } else {
                const double alphai = dri0 / curv;
#pragma unroll
                for (int j = 0; j < V::n; ++j) {
                    xs.v[j] += alphai * ps.v[j];
                    ri.v[j] += alphai * r.v[j];
                }
                const double dri1 = v_dot(ri, ri);
                ri_l1 = v_l1(ri);
                const double betai = dri1 / dri0;
#pragma unroll
                for (int j = 0; j < V::n; ++j) ps.v[j] = -ri.v[j] + betai * ps.v[j];
                ++cg_i;
                dri0 = dri1;
                ++cg_k2;
                phase = CG_TOP;
            }
        } else if (phase == WAIT_W1) {
            constexpr double ftol = 1e-4, gtol = 0.9, xtol = 1e-14, stpmin = 1e-8, stpmax = 50.0;
            constexpr double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
            const double finit = phi0, ginit = derphi0, gtest = u[GTEST];
            double stp = u[STP], stmin = u[STMIN], stmax = u[STMAX];
            const double g = v_dot(r, xs);
            const double ftest = finit + stp * gtest;
            if (!stage2 && f <= ftest && g >= 0) stage2 = true;
            int task = 0;  // 0 = continue, 1 = warning, 2 = convergence; a later test overrides an earlier one
            if (brackt && (stp <= stmin || stp >= stmax)) task = 1;
            if (brackt && stmax - stmin <= xtol * stmax) task = 1;
            if (stp == stpmax && f <= ftest && g <= gtest) task = 1;
            if (stp == stpmin && (f > ftest || g >= gtest)) task = 1;
            if (f <= ftest && fabs(g) <= gtol * -ginit) task = 2;
            if (task == 2) {
                accept(stp, f, r);
            } else if (task == 1) {
                phase = W2_START;
            } else {
                Step s;
                s.stx = u[STX]; s.fx = u[FX]; s.dx = u[DX]; s.sty = u[STY]; s.fy = u[FY]; s.dy = u[DY];
                s.stp = stp;
                s.brackt = brackt;
                if (!stage2 && f <= s.fx && f > ftest) {
                    // modified function psi(a) = phi(a) - phi(0) - ftol a phi'(0)
                    const double fm = f - stp * gtest;
                    Step m = s;
                    m.fx = s.fx - s.stx * gtest; m.fy = s.fy - s.sty * gtest;
                    m.dx = s.dx - gtest; m.dy = s.dy - gtest;
                    m_dcstep(m, fm, g - gtest, stmin, stmax);
                    s.stx = m.stx; s.sty = m.sty; s.stp = m.stp; s.brackt = m.brackt;
                    s.fx = m.fx + m.stx * gtest; s.fy = m.fy + m.sty * gtest;
                    s.dx = m.dx + gtest; s.dy = m.dy + gtest;
                } else {
                    m_dcstep(s, f, g, stmin, stmax);
                }
                stp = s.stp;
                double width = u[WIDTH], width1 = u[WIDTH1];
                if (s.brackt) {
                    if (fabs(s.sty - s.stx) >= p66 * width1) stp = s.stx + p5 * (s.sty - s.stx);
                    width1 = width;
                    width = fabs(s.sty - s.stx);
                }
                if (s.brackt) {
                    stmin = fmin(s.stx, s.sty);
                    stmax = fmax(s.stx, s.sty);
                } else {
                    stmin = stp + xtrapl * (stp - s.stx);
                    stmax = stp + xtrapu * (stp - s.stx);
                }
                stp = m_clamp(stp, stpmin, stpmax);
                if ((s.brackt && (stp <= stmin || stp >= stmax)) || (s.brackt && stmax - stmin <= xtol * stmax)) stp = s.stx;
                u[STX] = s.stx; u[FX] = s.fx; u[DX] = s.dx; u[STY] = s.sty; u[FY] = s.fy; u[DY] = s.dy;
                u[STMIN] = stmin; u[STMAX] = stmax; u[WIDTH] = width; u[WIDTH1] = width1; u[STP] = stp;
                brackt = s.brackt;
                ++it;
                phase = W1_TOP;
            }
        } else if (phase == WAIT_W2_FIRST) {
            u[PHI_A1] = f;
            u[PHI_A0] = phi0;
            u[DERPHI_A0] = derphi0;
            it = 0;
            phase = W2_TOP;
        } else if (phase == WAIT_W2) {
            u[PHI_A1] = f;
            ++it;
            phase = W2_TOP;
        } else if (phase == WAIT_ZOOM) {
            const double a_j = u[A_J], phi_aj = f;
            if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= u[PHI_LO]) {
                u[PHI_REC] = u[PHI_HI]; u[A_REC] = u[A_HI];
                u[A_HI] = a_j; u[PHI_HI] = phi_aj;
                ++it;
                phase = ZOOM_TOP;
            } else {
                const double derphi_aj = v_dot(r, xs);
                if (fabs(derphi_aj) <= -c2 * derphi0) {
                    accept(a_j, phi_aj, r);
                } else {
                    if (derphi_aj * (u[A_HI] - u[A_LO]) >= 0) {
                        u[PHI_REC] = u[PHI_HI]; u[A_REC] = u[A_HI];
                        u[A_HI] = u[A_LO]; u[PHI_HI] = u[PHI_LO];
                    } else {
                        u[PHI_REC] = u[PHI_LO]; u[A_REC] = u[A_LO];
                    }
                    u[A_LO] = a_j; u[PHI_LO] = phi_aj; u[DERPHI_LO] = derphi_aj;
                    ++it;
                    phase = ZOOM_TOP;
                }
            }
        }

        // transitions that need no evaluation, in the order of the common path (one pass of the loop); the bracketing
        // search's budget exit (W2_TOP -> ACCEPT) is the one edge that goes backwards
        while (phase >= FIRST_TRANSIENT) {
            if (phase == ACCEPT) {
                old_old_fval = old_fval;
                have_old_old = true;
                old_fval = new_fval;
                double l1 = 0.0;
#pragma unroll
                for (int j = 0; j < V::n; ++j) {
                    const double upd = m_mul(alphak, xs.v[j]);
                    xk.v[j] = m_point(xk.v[j], alphak, xs.v[j]);  // the bits of the accepted trial point
                    l1 += fabs(upd);
                }
                update_l1 = V::reduce(l1);
                ++k;
                phase = NEWTON_TOP;
            }
            if (phase == NEWTON_TOP) {
                if (!(update_l1 > dim * 1e-5)) {
                    status = OK;
                    phase = DONE;
                } else if (k >= maxiter) {
                    status = MAXITER;
                    phase = DONE;
                } else {
                    phase = CG_START;
                }
            }
            if (phase == CG_START) {
                // CG on  H p = -g  from p = 0; forcing term min(0.5, sqrt|g|_1)
                const double maggrad = v_l1(gv);
                const double eta = fmin(0.5, sqrt(maggrad));
                termcond = eta * maggrad;
#pragma unroll
                for (int j = 0; j < V::n; ++j) {
                    xs.v[j] = 0.0;
                    ri.v[j] = gv.v[j];
                    ps.v[j] = -gv.v[j];
                }
                cg_i = 0;
                cg_k2 = 0;
                dri0 = v_dot(ri, ri);
                ri_l1 = maggrad;  // ri = gv
                phase = CG_TOP;
            }
            if (phase == CG_TOP) {
                if (cg_k2 >= 20 * dim) {
                    status = CG_FAILED;
                    phase = DONE;
                } else if (ri_l1 <= termcond) {
                    phase = CG_END;
                } else {
                    phase = WAIT_CG_HP;
                }
            }
            if (phase == CG_END) {
                derphi0 = v_dot(gv, xs);
                phase = W1_START;
            }
            if (phase == W1_START) {
                constexpr double ftol = 1e-4, stpmin = 1e-8, stpmax = 50.0, p5 = 0.5, xtrapu = 4.0;
                double alpha1 = 1.0;
                if (have_old_old && derphi0 != 0.0) {
                    alpha1 = fmin(1.0, 1.01 * 2 * (old_fval - old_old_fval) / derphi0);
                    if (alpha1 < 0) alpha1 = 1.0;
                }
                if (alpha1 < stpmin || alpha1 > stpmax || !(derphi0 < 0)) {
                    phase = W2_START;  // the search's "ERROR" tasks
                } else {
                    brackt = false;
                    stage2 = false;
                    u[GTEST] = ftol * derphi0;
                    u[WIDTH] = stpmax - stpmin;
                    u[WIDTH1] = u[WIDTH] / p5;
                    u[STX] = 0.0; u[FX] = old_fval; u[DX] = derphi0;
                    u[STY] = 0.0; u[FY] = old_fval; u[DY] = derphi0;
                    u[STMIN] = 0.0;
                    u[STMAX] = alpha1 + xtrapu * alpha1;
                    u[STP] = alpha1;
                    it = 1;  // the START call consumed iteration 0 of the reference's loop
                    phase = W1_TOP;
                }
            }
            if (phase == W1_TOP) {
                if (it >= 100 || !isfinite(u[STP])) {
                    phase = W2_START;
                } else {
                    set_point(u[STP]);
                    phase = WAIT_W1;
                }
            }
            if (phase == W2_START) {
                double alpha1 = 1.0;
                if (have_old_old && derphi0 != 0.0) alpha1 = fmin(1.0, 1.01 * 2 * (old_fval - old_old_fval) / derphi0);
                if (alpha1 < 0) alpha1 = 1.0;
                u[ALPHA0] = 0.0;
                u[ALPHA1] = alpha1;
                set_point(alpha1);
                phase = WAIT_W2_FIRST;
            }
            if (phase == W2_TOP) {
                const double alpha0 = u[ALPHA0], alpha1 = u[ALPHA1], phi_a1 = u[PHI_A1], phi_a0 = u[PHI_A0];
                if (it >= 10) {
                    // budget exhausted: the reference accepts the last trial step (with a warning); its gradient is r
                    accept(alpha1, phi_a1, r);
                } else if (alpha1 == 0.0) {
                    status = LINESEARCH_FAILED;  // rounding errors prevent progress
                    phase = DONE;
                } else if (phi_a1 > old_fval + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && it > 0)) {
                    zoom_init(alpha0, alpha1, phi_a0, phi_a1, u[DERPHI_A0]);
                } else {
                    const double derphi_a1 = v_dot(r, xs);
                    if (fabs(derphi_a1) <= -c2 * derphi0) {
                        accept(alpha1, phi_a1, r);
                    } else if (derphi_a1 >= 0) {
                        zoom_init(alpha1, alpha0, phi_a1, phi_a0, derphi_a1);
                    } else {
                        u[ALPHA0] = alpha1;
                        u[ALPHA1] = 2 * alpha1;
                        u[PHI_A0] = phi_a1;
                        u[DERPHI_A0] = derphi_a1;
                        set_point(u[ALPHA1]);
                        phase = WAIT_W2;
                    }
                }
            }
            if (phase == ZOOM_TOP) {
                constexpr double delta1 = 0.2, delta2 = 0.1;
                if (it > 10) {
                    status = LINESEARCH_FAILED;
                    phase = DONE;
                } else {
                    const double a_lo = u[A_LO], a_hi = u[A_HI];
                    const double dalpha = a_hi - a_lo;
                    const double a = dalpha < 0 ? a_hi : a_lo, b = dalpha < 0 ? a_lo : a_hi;
                    double a_j = 0.0;
                    bool have = false;
                    if (it > 0) {
                        const double cchk = delta1 * dalpha;
                        have = m_cubicmin(a_lo, u[PHI_LO], u[DERPHI_LO], a_hi, u[PHI_HI], u[A_REC], u[PHI_REC], a_j);
                        if (have && (a_j > b - cchk || a_j < a + cchk)) have = false;
                    }
                    if (!have) {
                        const double qchk = delta2 * dalpha;
                        have = m_quadmin(a_lo, u[PHI_LO], u[DERPHI_LO], a_hi, u[PHI_HI], a_j);
                        if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
                    }
                    u[A_J] = a_j;
                    set_point(a_j);
                    phase = WAIT_ZOOM;
                }
            }
        }
    }
};

}  // namespace ncgm
}  // namespace salnmf
