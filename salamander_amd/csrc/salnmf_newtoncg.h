// Newton-CG with a strong-Wolfe line search, one wavefront per problem, fp64.
//
// The reference updates every CorrNMF embedding with
//   scipy.optimize.minimize(method="Newton-CG", jac=..., hess=...)   (_utils_corrnmf.py:400-407)
// i.e. the arithmetic of that step is SciPy's (pinned 1.13.1 in the reference's poetry.lock).  This
// header restates SciPy's published algorithm for that call -- the truncated Newton iteration of
// `_minimize_newtoncg` (inner conjugate-gradient solve with the min(0.5, sqrt|g|_1) forcing term),
// its line search `_line_search_wolfe12`: first the More-Thuente search of MINPACK-2 (`dcsrch` /
// `dcstep`, ftol 1e-4, gtol 0.9, xtol 1e-14, step bounds [1e-8, 50], <= 100 evaluations), and on
// failure the bracketing / zoom search of Nocedal & Wright (alg. 3.5 / 3.6 with the cubic / quadratic
// interpolation safeguards 0.2 / 0.1, <= 10 + 10 iterations) -- with the same constants, the same
// order of tests and the same termination rules, so that the iterates agree with SciPy's to rounding.
//
// Execution model: ONE wavefront solves ONE problem.  A vector of the problem (dimension <= 64) is
// one double per lane (lane m holds component m, lanes >= dim hold 0); every scalar of the algorithm
// is computed redundantly and bit-identically in all 64 lanes (wave_sum is a butterfly of commutative
// additions), so the control flow is wave-uniform and needs no divergence handling.  All loops are
// bounded by the constants above.
//
// The evaluator E supplies the problem:
//   double fun(double y)          objective at the point y (per-lane component) -> uniform scalar
//   double grad(double y)         per-lane component of the gradient at y
//   void   fun_grad(double y, double& f, double& g)   both at once (one pass over the problem's terms)
//   void   prepare_hess(double x) fix the Hessian at x
//   double hessp(double p)        per-lane component of (Hessian at the fixed point) . p
//   bool   exhausted()            true once an evaluation budget is spent (a guard against runaway solves;
//                                 uniform over the wave); the solve then stops with status MAXITER
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

namespace salnmf {
namespace ncg {

// Sum over the 64 lanes, the same bits in every lane.  Inside a row of 16 lanes by DPP moves (quad permutations, then the
// row's half mirror and mirror: partners hold equal values after the steps before, so a + b and b + a meet), across the
// four rows by v_permlane16_swap / v_permlane32_swap of the value with itself: no LDS round trip.  As a butterfly of
// __shfl_xor (two ds_bpermute per step on doubles) a sum cost ~600 cycles, and a CG iteration takes three of them.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double dot(double a, double b) { return wave_sum(a * b); }
// x + t * p with the product rounded first (no fused multiply-add): the trial points of a line search and the
// accepted iterate must be the same bits, as they are in NumPy, so that a gradient evaluated by the line search
// can be reused for the next Newton iteration (SciPy's ScalarFunction memoises exactly this evaluation)
__device__ __forceinline__ double point(double x, double t, double p) { return __dadd_rn(x, __dmul_rn(t, p)); }
__device__ __forceinline__ double l1norm(double a) { return wave_sum(fabs(a)); }

enum Status { OK = 0, MAXITER = 1, LINESEARCH_FAILED = 2, CG_FAILED = 3 };

// ---- MINPACK-2 dcstep: safeguarded cubic / quadratic step inside (or extending) the bracket
struct StepState {
    double stx, fx, dx, sty, fy, dy, stp;
    bool brackt;
};

__device__ inline double sign_of(double v) { return (v > 0.0) - (v < 0.0); }
__device__ inline double clamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ inline void dcstep(StepState& s, double fp, double dp, double stpmin, double stpmax) {
    const double sgnd = sign_of(dp) * sign_of(s.dx);
    double stpf;
    const double stp = s.stp;
    if (fp > s.fx) {
        // higher function value: the minimum is bracketed
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
        // lower value, derivatives of opposite sign: bracketed
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
        // lower value, same sign, derivative magnitude decreases
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
            stpf = clamp(stpf, stpmin, stpmax);
        }
    } else {
        // lower value, same sign, derivative magnitude does not decrease
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
    // update the interval that contains a minimiser
    if (fp > s.fx) {
        s.sty = stp; s.fy = fp; s.dy = dp;
    } else {
        if (sgnd < 0.0) { s.sty = s.stx; s.fy = s.fx; s.dy = s.dx; }
        s.stx = stp; s.fx = fp; s.dx = dp;
    }
    s.stp = stpf;
}

// ---- More-Thuente search (MINPACK-2 dcsrch as driven by scipy's scalar_search_wolfe1)
// phi(alpha) = f(xk + alpha pk).  Returns true and the accepted step / values on convergence.
template <class E>
__device__ inline bool search_wolfe1(E& ev, double xk, double pk, double phi0, bool have_old, double old_phi0, double derphi0,
                                     double& stp_out, double& phi_out, double& grad_out) {
    constexpr double ftol = 1e-4, gtol = 0.9, xtol = 1e-14, stpmin = 1e-8, stpmax = 50.0;
    constexpr double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    double alpha1 = 1.0;
    if (have_old && derphi0 != 0.0) {
        alpha1 = fmin(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
        if (alpha1 < 0) alpha1 = 1.0;
    }
    // START
    if (alpha1 < stpmin || alpha1 > stpmax || !(derphi0 < 0)) return false;  // "ERROR" tasks
    StepState s;
    s.brackt = false;
    int stage = 1;
    const double finit = phi0, ginit = derphi0, gtest = ftol * ginit;
    double width = stpmax - stpmin, width1 = width / p5;
    s.stx = 0.0; s.fx = finit; s.dx = ginit;
    s.sty = 0.0; s.fy = finit; s.dy = ginit;
    double stmin = 0.0, stmax = alpha1 + xtrapu * alpha1;
    double stp = alpha1;
    // the START call consumed iteration 0 of the reference's loop; each later one evaluates then iterates
    for (int it = 1; it < 100; ++it) {
        if (!isfinite(stp)) return false;
        double f, gvec;
        ev.fun_grad(point(xk, stp, pk), f, gvec);
        const double g = dot(gvec, pk);
        const double ftest = finit + stp * gtest;
        if (stage == 1 && f <= ftest && g >= 0) stage = 2;
        // tests in the reference's order: a later one overrides an earlier one
        int task = 0;  // 0 = continue, 1 = warning, 2 = convergence
        if (s.brackt && (stp <= stmin || stp >= stmax)) task = 1;
        if (s.brackt && stmax - stmin <= xtol * stmax) task = 1;
        if (stp == stpmax && f <= ftest && g <= gtest) task = 1;
        if (stp == stpmin && (f > ftest || g >= gtest)) task = 1;
        if (f <= ftest && fabs(g) <= gtol * -ginit) task = 2;
        if (task == 2) { stp_out = stp; phi_out = f; grad_out = gvec; return true; }
        if (task == 1) return false;
        s.stp = stp;
        if (stage == 1 && f <= s.fx && f > ftest) {
            // modified function psi(a) = phi(a) - phi(0) - ftol a phi'(0)
            const double fm = f - stp * gtest;
            StepState m = s;
            m.fx = s.fx - s.stx * gtest; m.fy = s.fy - s.sty * gtest;
            m.dx = s.dx - gtest; m.dy = s.dy - gtest;
            dcstep(m, fm, g - gtest, stmin, stmax);
            s.stx = m.stx; s.sty = m.sty; s.stp = m.stp; s.brackt = m.brackt;
            s.fx = m.fx + m.stx * gtest; s.fy = m.fy + m.sty * gtest;
            s.dx = m.dx + gtest; s.dy = m.dy + gtest;
        } else {
            dcstep(s, f, g, stmin, stmax);
        }
        stp = s.stp;
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
        stp = clamp(stp, stpmin, stpmax);
        if ((s.brackt && (stp <= stmin || stp >= stmax)) || (s.brackt && stmax - stmin <= xtol * stmax)) stp = s.stx;
    }
    return false;  // did not converge within the evaluation budget
}

// ---- interpolation helpers of the zoom phase; false = "no usable minimiser"
__device__ inline bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double& xmin) {
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
__device__ inline bool quadmin(double a, double fa, double fpa, double b, double fb, double& xmin) {
    const double D = fa, C = fpa, db = b - a;
    if (db * db == 0.0) return false;
    const double B = (fb - D - C * db) / (db * db);
    if (2.0 * B == 0.0) return false;
    xmin = a - C / (2.0 * B);
    return isfinite(xmin);
}

template <class E>
__device__ inline bool zoom(E& ev, double xk, double pk, double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo,
                            double phi0, double derphi0, double& a_star, double& val_star, double& grad_out) {
    constexpr double c1 = 1e-4, c2 = 0.9, delta1 = 0.2, delta2 = 0.1;
    double phi_rec = phi0, a_rec = 0.0;
    for (int i = 0; i <= 10; ++i) {
        const double dalpha = a_hi - a_lo;
        const double a = dalpha < 0 ? a_hi : a_lo, b = dalpha < 0 ? a_lo : a_hi;
        double a_j = 0.0;
        bool have = false;
        if (i > 0) {
            const double cchk = delta1 * dalpha;
            have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j);
            if (have && (a_j > b - cchk || a_j < a + cchk)) have = false;
        }
        if (!have) {
            const double qchk = delta2 * dalpha;
            have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j);
            if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
        }
        const double phi_aj = ev.fun(point(xk, a_j, pk));
        if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
            phi_rec = phi_hi; a_rec = a_hi;
            a_hi = a_j; phi_hi = phi_aj;
        } else {
            const double gvec = ev.grad(point(xk, a_j, pk));
            const double derphi_aj = dot(gvec, pk);
            if (fabs(derphi_aj) <= -c2 * derphi0) { a_star = a_j; val_star = phi_aj; grad_out = gvec; return true; }
            if (derphi_aj * (a_hi - a_lo) >= 0) {
                phi_rec = phi_hi; a_rec = a_hi;
                a_hi = a_lo; phi_hi = phi_lo;
            } else {
                phi_rec = phi_lo; a_rec = a_lo;
            }
            a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
        }
    }
    return false;
}

// ---- bracketing search (Nocedal & Wright alg. 3.5) used when the More-Thuente search gives up
template <class E>
__device__ inline bool search_wolfe2(E& ev, double xk, double pk, double phi0, bool have_old, double old_phi0, double derphi0,
                                     double& stp_out, double& phi_out, double& grad_out, bool& have_grad) {
    constexpr double c1 = 1e-4, c2 = 0.9;
    double alpha0 = 0.0, alpha1 = 1.0;
    if (have_old && derphi0 != 0.0) alpha1 = fmin(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
    if (alpha1 < 0) alpha1 = 1.0;
    double phi_a1 = ev.fun(point(xk, alpha1, pk)), phi_a0 = phi0, derphi_a0 = derphi0;
    have_grad = true;
    for (int i = 0; i < 10; ++i) {
        if (alpha1 == 0.0) return false;  // rounding errors prevent progress
        if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && i > 0))
            return zoom(ev, xk, pk, alpha0, alpha1, phi_a0, phi_a1, derphi_a0, phi0, derphi0, stp_out, phi_out, grad_out);
        const double gvec = ev.grad(point(xk, alpha1, pk));
        const double derphi_a1 = dot(gvec, pk);
        if (fabs(derphi_a1) <= -c2 * derphi0) { stp_out = alpha1; phi_out = phi_a1; grad_out = gvec; return true; }
        if (derphi_a1 >= 0)
            return zoom(ev, xk, pk, alpha1, alpha0, phi_a1, phi_a0, derphi_a1, phi0, derphi0, stp_out, phi_out, grad_out);
        alpha0 = alpha1;
        alpha1 = 2 * alpha1;
        phi_a0 = phi_a1;
        phi_a1 = ev.fun(point(xk, alpha1, pk));
        derphi_a0 = derphi_a1;
    }
    // budget exhausted: the reference accepts the last trial step (with a warning); no gradient at that point yet
    stp_out = alpha1;
    phi_out = phi_a1;
    have_grad = false;
    return true;
}

// The state of a solve at the top of a Newton iteration: everything the loop carries.  A caller that runs minimize again
// and again on a growing record of evaluations (the lockstep signature solves) passes one, and the solve resumes at the
// last iteration it had reached instead of at the start point -- the same arithmetic from there on.
struct NoCheckpoint {
    static constexpr bool enabled = false;
};
struct Checkpoint {
    static constexpr bool enabled = true;
    double xk, g_next;  // per lane
    double old_fval, old_old_fval, update_l1norm;
    int k, have_old_old, have_g_next;
    int tag;            // the evaluator's own position (E::tag / E::set_tag), e.g. a cursor into its record
    bool valid;
};

// ---- truncated Newton iteration.  x: per-lane component of the start, overwritten by the result.
template <class E, class CP = NoCheckpoint>
__device__ inline int minimize(E& ev, double& x, int dim, int maxiter, int* n_iter = nullptr, CP* cp = nullptr) {
    const double xtol = dim * 1e-5;
    const int cg_maxiter = 20 * dim;
    constexpr double float64eps = DBL_EPSILON;
    double update_l1norm = DBL_MAX;
    double xk = x;
    int k = 0;
    double old_fval, old_old_fval = 0.0;
    bool have_old_old = false;
    int status = OK;
    double g_next;  // gradient at xk when an evaluation at that very point already produced it
    bool have_g_next = true;
    bool resumed = false;
    if constexpr (CP::enabled) {
        if (cp->valid) {
            xk = cp->xk, g_next = cp->g_next, old_fval = cp->old_fval, old_old_fval = cp->old_old_fval, update_l1norm = cp->update_l1norm;
            k = cp->k, have_old_old = cp->have_old_old != 0, have_g_next = cp->have_g_next != 0;
            ev.set_tag(cp->tag);
            resumed = true;
        }
    }
    if (!resumed) ev.fun_grad(xk, old_fval, g_next);
    while (update_l1norm > xtol) {
        if (k >= maxiter || ev.exhausted()) { status = MAXITER; break; }
        if constexpr (CP::enabled) {  // (not exhausted: every value below is the solve's own)
            cp->xk = xk, cp->g_next = g_next, cp->old_fval = old_fval, cp->old_old_fval = old_old_fval, cp->update_l1norm = update_l1norm;
            cp->k = k, cp->have_old_old = have_old_old, cp->have_g_next = have_g_next, cp->tag = ev.tag(), cp->valid = true;
        }
        // search direction: CG on  H p = -g  from p = 0, stopped by the forcing term or by curvature
        const double gfk = have_g_next ? g_next : ev.grad(xk);
        const double b = -gfk;
        const double maggrad = l1norm(b);
        const double eta = fmin(0.5, sqrt(maggrad));
        const double termcond = eta * maggrad;
        double xsupi = 0.0, ri = -b, psupi = -ri;
        int i = 0;
        double dri0 = dot(ri, ri);
        ev.prepare_hess(xk);
        bool cg_done = false;
        for (int k2 = 0; k2 < cg_maxiter; ++k2) {
            if (l1norm(ri) <= termcond || ev.exhausted()) { cg_done = true; break; }
            const double Ap = ev.hessp(psupi);
            const double curv = dot(psupi, Ap);
            if (0 <= curv && curv <= 3 * float64eps) { cg_done = true; break; }
            if (curv < 0) {
                if (i == 0) xsupi = dri0 / (-curv) * b;  // steepest descent fallback
                cg_done = true;
                break;
            }
            const double alphai = dri0 / curv;
            xsupi += alphai * psupi;
            ri += alphai * Ap;
            const double dri1 = dot(ri, ri);
            const double betai = dri1 / dri0;
            psupi = -ri + betai * psupi;
            ++i;
            dri0 = dri1;
        }
        if (!cg_done) { status = CG_FAILED; break; }
        const double pk = xsupi;
        const double derphi0 = dot(gfk, pk);
        double alphak = 0.0, new_fval = 0.0;
        have_g_next = true;
        bool ok = search_wolfe1(ev, xk, pk, old_fval, have_old_old, old_old_fval, derphi0, alphak, new_fval, g_next);
        if (!ok) ok = search_wolfe2(ev, xk, pk, old_fval, have_old_old, old_old_fval, derphi0, alphak, new_fval, g_next, have_g_next);
        if (!ok) { status = LINESEARCH_FAILED; break; }
        old_old_fval = old_fval;
        have_old_old = true;
        old_fval = new_fval;
        const double update = __dmul_rn(alphak, pk);
        xk = __dadd_rn(xk, update);  // the same bits as the accepted trial point of the line search
        ++k;
        update_l1norm = l1norm(update);
    }
    x = xk;
    if (n_iter) *n_iter = k;
    return status;
}

}  // namespace ncg
}  // namespace salnmf
