// rfgpu_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the RF_INV
// forward + likelihood hot path.  All arithmetic is IEEE fp64 on the vector ALU.
//
// Reference behaviour restated (file:line under /root/reference):
//   K1 spectra_kernel   src/forward.f90:212-287 (calc_seis) with :350-380 (e_inverse),
//                       :385-421 (layer_matrix_sol), :424-442 (layer_matrix_liq) and the
//                       conj / -conj of :145-146
//   K2 trace_kernel     src/forward.f90:148-203 (water_level_decon :447-470, direct_arrival
//                       :474-519, filter, c2r, time shift, vertical normalisation) and
//                       src/likelihood.f90:87-93 (misfit, misfit . R^-1 . misfit)
//   K3 logl_kernel      src/likelihood.f90:86,94-96
//
// Formulation of K1: every solid-layer propagator P has the
// checkerboard pattern "real where i+j even, imaginary otherwise" and an explicit
// omega / 1/omega in its off-diagonal 2x2 blocks, i.e.  P = T A T^-1 with
// T = diag(1, i, w, i w) and A REAL with no explicit omega.  The chain
// prod P_l = T (prod A_l) T^-1 is therefore a chain of real 4x4 products, E^-1 T is
// omega-independent, and only the columns of the product that the boundary condition
// consumes (1, 2 and, under an ocean, 4) are propagated.
#include "rfgpu_internal.h"
#include <math.h>
#include <atomic>
#include <type_traits>

// Timing diagnostics (tools/ablate.sh): a build with -DRFGPU_DIAGNOSTICS can stop a block after phase N
// to split the kernel time; the production library has no such exits.
#ifdef RFGPU_DIAGNOSTICS
#define RFGPU_ABLATE_AT(n, ret) do { if (P.ablate == (n)) return ret; } while (0)
#define RFGPU_ABLATE_DO(n, stmt) if (P.ablate == (n)) stmt
#define RFGPU_ABLATE_IS(n) (P.ablate == (n))
#define RFGPU_ABLATE_COPY(dst) (dst).ablate = P.ablate
// "ablate" 100 + X (results stay VALID): every other block -- by a hash of its index -- idles X microseconds before it
// starts, so that the blocks sharing a CU run out of phase (one propagates while the other transforms): the experiment of
// profiles/EXPERIMENTS.md round 6.  1000 + X: only blocks of the first 512 (the first round of resident blocks).
#define RFGPU_STAGGER(bid)                                                                                       \
    do {                                                                                                         \
        const int st_ = P.ablate >= 1000 ? P.ablate - 1000 : P.ablate - 100;                                     \
        if (P.ablate >= 100 && (P.ablate < 1000 || (bid) < 512) && (((unsigned)(bid) * 2654435761u) >> 16 & 1u)) \
            for (int i_ = 0; i_ < st_; ++i_) __builtin_amdgcn_s_sleep(32);   /* ~1 us each */                                   \
    } while (0)
#else
#define RFGPU_STAGGER(bid) do { } while (0)
#define RFGPU_ABLATE_AT(n, ret) do { } while (0)
#define RFGPU_ABLATE_DO(n, stmt)
#define RFGPU_ABLATE_IS(n) false
#define RFGPU_ABLATE_COPY(dst)
#endif

namespace rfgpu {

// Dynamic LDS beyond 64 KiB must be opted into per kernel and per device.  Every kernel that can need it is
// opted in ONCE to the CU's whole 160 KiB (the launch's own size decides residency, not this ceiling): the
// same value from every thread, so concurrent contexts on different host threads cannot shrink each other's
// limit, and nothing is set on the launch path afterwards.
struct LdsOptIn {
    std::atomic<unsigned> done{0};   // bit per device ordinal (mod 32)
    void operator()(const void *fn)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (done.load(std::memory_order_acquire) & bit) return;
        // the ceiling is the CU's LDS minus the kernel's static allocation
        hipFuncAttributes attr{};
        size_t stat = 0;
        if (hipFuncGetAttributes(&attr, fn) == hipSuccess) stat = attr.sharedSizeBytes;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - stat)) != hipSuccess)
            (void)hipGetLastError();   // leave no sticky error behind: a launch that needs the room reports it itself
        done.fetch_or(bit, std::memory_order_release);
    }
};

// ---------------------------------------------------------------------------
// small complex helpers (same operation order as the oracle's c_mul)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 cneg(double2 a) { return make_double2(-a.x, -a.y); }
// Two quotients over one denominator (the boundary conditions divide both displacement components
// by the same determinant).  Fortran compilers emit Smith's range-reduced division for complex(8)
// a / b (the oracle's c_div): |b.re| >= |b.im| ? r = b.im / b.re, d = b.re + b.im r,
// ((a.re + a.im r) / d, (a.im - a.re r) / d) : r = b.re / b.im, d = b.im + b.re r,
// ((a.re r + a.im) / d, (a.im r - a.re) / d).  Here: the same branch choice and numerator forms,
// branch-free, the reduction shared by both quotients and one reciprocal instead of four divisions
// by d (each quotient <= 1.5 ulp from Smith's).
__device__ __forceinline__ void cdiv2(double2 a1, double2 a2, double2 b, double2 &x1, double2 &x2)
{
    const bool sel = fabs(b.x) >= fabs(b.y);
    const double p = sel ? b.x : b.y, q = sel ? b.y : b.x;
    const double r = q / p;
    const double t = 1.0 / (p + q * r);
    x1 = make_double2((sel ? a1.x + a1.y * r : a1.x * r + a1.y) * t, (sel ? a1.y - a1.x * r : a1.y * r - a1.x) * t);
    x2 = make_double2((sel ? a2.x + a2.y * r : a2.x * r + a2.y) * t, (sel ? a2.y - a2.x * r : a2.y * r - a2.x) * t);
}

// The fast paths' form of the same two quotients: a conj(b) / |b|^2 with one reciprocal -- 27 instructions instead of
// 51, no selects.  Smith's range reduction protects |b| beyond 1e154 or below 1e-154, which the boundary-condition
// determinant of a walker on the fast paths (unit gauge, phases below 1e6 rad) never reaches; the generic path keeps
// cdiv2.  Each quotient within 3 ulp of the exact one (Smith: 2.5).
//
// Round 4 ("one quotient"): ONE reciprocal of the determinant, w = conj(b) / |b|^2, and both numerators
// times w -- 2 + 4 + 4 multiply-adds instead of 2 x (4 + 2) -- with 1 / |b|^2 from v_rcp_f64 and two Newton steps
// (5 instructions, ~1 ulp) instead of the IEEE division's scale / fmas / fixup sequence (12): that sequence exists for
// the subnormal and overflow ranges, which |b|^2 of a walker on the fast paths never reaches (see above); 0, Inf and
// NaN still give Inf / 0 / NaN (a Newton step of an exact Inf or 0 would give NaN: the selects keep the seed).
__device__ __forceinline__ double rcp_nr2(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e0 = fma(-x, r0, 1.0);
    const double r1 = fma(r0, e0, r0);
    const double e1 = fma(-x, r1, 1.0);
    const double r2 = fma(r1, e1, r1);
    return (e0 == e0 && fabs(r0) < HUGE_VAL) ? r2 : r0;     // x = 0, Inf, NaN: the seed is already the answer
}

__device__ __forceinline__ void cdiv2_norm(double2 a1, double2 a2, double2 b, double2 &x1, double2 &x2)
{
    const double t = rcp_nr2(fma(b.x, b.x, b.y * b.y));
    const double wx = b.x * t, wy = -(b.y * t);             // w = conj(b) / |b|^2
    x1 = make_double2(fma(a1.x, wx, -(a1.y * wy)), fma(a1.x, wy, a1.y * wx));
    x2 = make_double2(fma(a2.x, wx, -(a2.y * wy)), fma(a2.x, wy, a2.y * wx));
}

// direct_arrival (forward.f90:474-519).  Its result feeds nint() (integer bookkeeping
// must be bit-exact), so no FMA contraction anywhere and the SUM stays strictly
// sequential in layer order; only the independent per-layer terms h(i)*sqrt(1/v(i)^2-p^2)
// are evaluated by separate lanes (each one the same IEEE operations as the reference).
__device__ __noinline__ double arrival_term(double h, double v, double rayp)
{
#pragma clang fp contract(off)
    const double vv = v * v;
    const double inv = 1.0 / vv;
    const double pp = rayp * rayp;
    const double rad = inv - pp;
    return h * sqrt(rad);
}

__device__ __noinline__ double arrival_sum(int n, const double *terms)
{
#pragma clang fp contract(off)
    double t = 0.0;
    for (int i = 0; i < n; ++i) t = t + terms[i];
    return t;
}

// ---------------------------------------------------------------------------
// K1  spectra
// ---------------------------------------------------------------------------
// Formulation.  Every solid-layer propagator of the reference (forward.f90:403-418) is
// P = T A T^-1 with T = diag(1, i, w, i w) and A(h) real and free of explicit w, and
// A = U R U^-1: R rotates a P pair (a_p, b_p) by the layer's P phase w xi h and an S pair
// (a_s, b_s) by its S phase w eta h; U (layer_basis below) is constant per layer.  The chain
// E^-1 P_n ... P_1 is therefore
//     (E^-1 T U_n) R_n (U_n^-1 U_{n-1}) R_{n-1} ... R_1 (U_1^-1 e_j):
// per (bin, layer) and propagated column 2 rotations (8 flop-instructions) and one change of
// eigen-coordinates G = U_next^-1 U_this, which couples only (a_p, b_s) and (b_p, a_s) -- two 2x2
// blocks, 8 instructions, wave-uniform coefficients.  The phases are the reference's own doubles
// (w xi) h.  Only the columns the boundary condition consumes are propagated (1, 2; ocean: + 4).
//
// Constants per (batch item, forward-trace), written once per batch item by stage_kernel (K0) into a global image
// (gcoef / gtail of WalkerState).  The fast paths read it through the scalar data path (KPtr); a walker on the
// generic path copies it into LDS (load_staged).
//   coef[l][0..NCOEF-1]  per solid layer l (0-based, l < nlay-1):
//      0 xi   1 eta   2 h
//      3..10  G: a_p'<-(a_p, b_s), b_s'<-(a_p, b_s), b_p'<-(b_p, a_s), a_s'<-(b_p, a_s)
//             (identity below the last solid layer; in the unit gauge c[3] = c[10] = 1, see stage_interface)
//     11,12 phi_xi  = domg*xi*h  as a double-double (hi, lo)   } phase per bin of the layer and
//     13,14 phi_eta = domg*eta*h as a double-double            } cos/sin of 64 bins of phase:
//     15,16 cos, sin(64 phi_xi)   17,18 cos, sin(64 phi_eta)   } the chained-phase path (below)
//     19,20 sin, cos of the layer's P phase at the Nyquist bin   } that bin has an iteration of its own
//     21,22 sin, cos of the layer's S phase at the Nyquist bin   } (one active lane): spectra_iter_nyquist
//     23 pad
//   tail[0..7]   rows 3,4 of E^-1 T of the half-space in the last solid layer's eigen-coordinates
//   tail[8..10]  water layer: xi_w, h_w, rho_w / xi_w
//   tail[11..16] unit columns 1, 2, 4 in the top solid layer's eigen-coordinates (stage_start)
//   tail[17]     direct-arrival time of the forward trace (forward.f90:474-519)
//   tail[18..21] (ur, uz) of the Nyquist bin when that bin sits alone in its 64-bin iteration (nfft a multiple of
//                128) and the walker is on the fast paths: stage_kernel runs that one bin's chain itself (stage_nyquist)
__device__ __forceinline__ void sincos_cw(double x, double &sn, double &cs);

// vertical slowness sqrt(1/v^2 - p^2) exactly as the reference's double arithmetic forms it
// (forward.f90:394-395: no FMA on the reference's x86-64 build); it enters the phase argument
// (omega * slowness) * h, whose rounding is part of the reference result
__device__ __forceinline__ double vertical_slowness(double v, double p)
{
#pragma clang fp contract(off)
    const double inv = 1.0 / (v * v);
    const double p2 = p * p;
    return sqrt(inv - p2);
}

__device__ __forceinline__ void stage_phase(double *c4, double *cs2, double domg, double slow, double h)
{
    // phi = domg * slow * h as hi + lo (error-free products), then cos/sin(64 phi)
    const double t_hi = domg * slow;
    const double t_lo = fma(domg, slow, -t_hi);
    const double ph = t_hi * h;
    double pl = fma(t_hi, h, -ph);
    pl = fma(t_lo, h, pl);
    c4[0] = ph;
    c4[1] = pl;
    double s, c;
    sincos_cw(64.0 * ph, s, c);
    const double d = 64.0 * pl;
    cs2[0] = fma(-s, d, c);
    cs2[1] = fma(c, d, s);
}

// Eigen-coordinates of a solid layer (see the header comment): x = U q with q = (a_p, b_p, a_s, b_s),
//   x1 = p a_p + eta b_s          x4 = rho bp a_p - 2 b^2 rho p eta b_s        (support {1,4})
//   x2 = xi b_p + p a_s           x3 = -2 b^2 rho p xi b_p + rho bp a_s        (support {2,3})
// and q = W^T x,
//   a_p = 2 b^2 p x1 + x4 / rho   b_s = (bp x1 - (p / rho) x4) / eta
//   a_s = 2 b^2 p x2 + x3 / rho   b_p = (bp x2 - (p / rho) x3) / xi
struct LayerBasis {
    double p, xi, eta, rho, bp, tb2p;   // tb2p = 2 beta^2 p
};

__device__ __forceinline__ LayerBasis layer_basis(double alpha, double beta, double rho, double p)
{
    LayerBasis b;
    const double b2 = beta * beta, p2 = p * p;
    b.p = p;
    b.rho = rho;
    b.bp = 1.0 - 2.0 * b2 * p2;
    b.eta = vertical_slowness(beta, p);
    b.xi = vertical_slowness(alpha, p);
    b.tb2p = 2.0 * b2 * p;
    return b;
}

// c[3..10] = G = W_next^T U_this: the change of eigen-coordinates across the interface below the
// layer (two 2x2 blocks: (a_p, b_s) and (b_p, a_s)); identity after the last solid layer.  Staged in
// four independent parts so that the lanes of stage_kernel's wave share the latency:
//   part 0: the (a_p, b_s) block, c[3..6]   (needs the S slownesses only)
//   part 1: the (b_p, a_s) block, c[7..10]  (needs the P slownesses only)
//   part 2: xi, h and the P phase constants;  part 3: eta and the S phase constants
struct LayerHalf {
    double p, rho, bp, tb2p, slow;   // slow = eta (part 0) or xi (part 1)
};

__device__ __forceinline__ LayerHalf layer_half(double vslow, double beta, double rho, double p)
{
    LayerHalf b;
    const double b2 = beta * beta, p2 = p * p;
    b.p = p;
    b.rho = rho;
    b.bp = 1.0 - 2.0 * b2 * p2;
    b.tb2p = 2.0 * b2 * p;
    b.slow = vertical_slowness(vslow, p);
    return b;
}

// part 0 (slow = eta) writes c[3..6]; part 1 (slow = xi) writes c[7..10]; the two blocks have the
// same entries up to their order: block 0 = (a_p'<-a_p, a_p'<-b_s, b_s'<-a_p, b_s'<-b_s),
// block 1 = (b_p'<-b_p, b_p'<-a_s, a_s'<-b_p, a_s'<-a_s)
//
// Gauge.  Scaling all four eigen-coordinates of a layer by one number leaves the rotations alone and rescales G;
// the entry a_p'<-a_p of block 0 and the entry a_s'<-a_s of block 1 are the SAME number d (interface_diag: it
// does not contain a vertical slowness), so dividing G by d makes both exactly 1 and saves two multiplications
// per propagated column, bin and layer in the chained-phase loop (apply_layer_trig_unit).  The product of the
// d's of a walker is folded into the half-space rows (stage_halfspace), so the boundary condition sees the
// unscaled product.  `unit` is decided per walker (walker_gauge): every d within [1/16, 16] and their product
// within 2^+-600; otherwise the walker keeps the plain G and takes the generic path.
__device__ __forceinline__ double interface_diag(const LayerHalf &u, const LayerHalf &w)
{
    return fma(w.tb2p, u.p, (u.rho / w.rho) * u.bp);
}

__device__ __forceinline__ void stage_interface(double *g, int part, const LayerHalf &u, const LayerHalf *w, bool unit)
{
    if (!w) {
        g[0] = 1.0; g[1] = 0.0; g[2] = 0.0; g[3] = 1.0;
        return;
    }
    const double pr = w->p / w->rho;            // p / rho'
    const double m = u.tb2p * u.rho;            // 2 b^2 rho p
    double diag_a = interface_diag(u, *w);
    // the unit gauge's 1 / d rides on the two divisors the entries have anyway (no division of its own)
    const double ga = unit ? diag_a : 1.0;
    double off_a = u.slow * (w->tb2p - m / w->rho) / ga;
    const double ws = w->slow * ga;
    double off_b = (w->bp * u.p - pr * (u.rho * u.bp)) / ws;
    double diag_b = u.slow * (w->bp + pr * m) / ws;
    if (unit) diag_a = 1.0;
    if (part == 0) {
        g[0] = diag_a; g[1] = off_a; g[2] = off_b; g[3] = diag_b;
    } else {
        g[0] = diag_b; g[1] = off_b; g[2] = off_a; g[3] = diag_a;
    }
}

// The walker's gauge, by the G lanes (G = 16, 32 or 64, aligned) that stage one (item, trace) -- every one of them
// returns the same values: scale = product of interface_diag over the interfaces between solid layers
// ilay0 .. nl-2, unit = the gauge is usable (see above).  L: the walker's layer rows [4][pad]; sl: lane within
// the group; group_mask: the group's lanes within the wave.
template <int G>
__device__ __forceinline__ double walker_gauge(const double *L, int pad, int nl, int ilay0, double p, int sl,
                                               unsigned long long group_mask, bool &unit)
{
    double prod = 1.0;
    bool ok = true;
    for (int l0 = ilay0; l0 + 1 < nl - 1; l0 += G) {
        const int l = l0 + sl;
        double d = 1.0;
        if (l + 1 < nl - 1) {
            const LayerHalf u = layer_half(L[pad + l], L[pad + l], L[2 * pad + l], p);
            const LayerHalf w = layer_half(L[pad + l + 1], L[pad + l + 1], L[2 * pad + l + 1], p);
            d = interface_diag(u, w);
            ok = ok && fabs(d) >= 0.0625 && fabs(d) <= 16.0;      // (false for NaN)
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) d *= __shfl_xor(d, o, 64);
        prod *= d;
    }
    const bool all_ok = (__ballot(ok) & group_mask) == (__ballot(true) & group_mask);
    unit = all_ok && fabs(prod) > 0x1p-600 && fabs(prod) < 0x1p600;
    return unit ? prod : 1.0;
}

// tail[0..7]: rows 3 and 4 of E^-1 (forward.f90:370-377) times T = diag(1, i, w, i w), expressed in
// the eigen-coordinates of the last solid layer (`last`; nullptr: no solid layer, physical
// coordinates).  Row r: re = g[0] v[0] + g[1] v[3], im = g[2] v[1] + g[3] v[2] (halfspace_row).
__device__ __forceinline__ void stage_halfspace(double *g, double alpha, double beta, double rho, double p,
                                                const LayerBasis *last, double scale)
{
    const double eta = vertical_slowness(beta, p);
    const double xi = vertical_slowness(alpha, p);
    const double bp = 1.0 - 2.0 * beta * beta * p * p;
    double e[8];
    e[0] = beta * beta * p / alpha;        // re: x1
    e[1] = 1.0 / (2.0 * rho * alpha);      // re: x4
    e[2] = -bp / (2.0 * alpha * xi);       // im: x2
    e[3] = p / (2.0 * rho * alpha * xi);   // im: x3
    e[4] = bp / (2.0 * beta * eta);        // re: x1
    e[5] = -p / (2.0 * rho * beta * eta);  // re: x4
    e[6] = beta * p;                       // im: x2
    e[7] = 1.0 / (2.0 * rho * beta);       // im: x3
    for (int r = 0; r < 2; ++r) {
        const double *q = e + 4 * r;
        double *o = g + 4 * r;
        if (last) {
            // (scale: the walker's gauge, 1 when the plain G is staged)
            const double m = last->tb2p * last->rho, rb = last->rho * last->bp;
            o[0] = scale * fma(q[0], last->p, q[1] * rb);       // a_p
            o[1] = scale * (last->eta * (q[0] - q[1] * m));     // b_s
            o[2] = scale * (last->xi * (q[2] - q[3] * m));      // b_p
            o[3] = scale * fma(q[2], last->p, q[3] * rb);       // a_s
        } else {
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
        }
    }
}

// tail[11..16]: the unit columns 1, 2 (and 4, ocean) of the product in the eigen-coordinates of the
// top solid layer: e1 -> (a_p, b_s) = (t[0], t[1]); e2 -> (b_p, a_s) = (t[2], t[3]);
// e4 -> (a_p, b_s) = (t[4], t[5])
__device__ __forceinline__ void stage_start(double *t, const LayerBasis *top)
{
    if (top) {
        t[0] = top->tb2p;
        t[1] = top->bp / top->eta;
        t[2] = top->bp / top->xi;
        t[3] = top->tb2p;
        t[4] = 1.0 / top->rho;
        t[5] = -(top->p / top->rho) / top->eta;
    } else {
        t[0] = 1.0; t[1] = 0.0; t[2] = 1.0; t[3] = 0.0; t[4] = 0.0; t[5] = 1.0;
    }
}

// ---------------------------------------------------------------------------
// sincos for the propagator phases.  The reference calls libm sin/cos on the double
// (omega*xi)*z (forward.f90:397-400); arguments reach a few hundred radians.  ocml's
// sincos reduces EVERY argument with the Payne-Hanek v_trig_preop path (~150
// instructions); this one uses a 3-term Cody-Waite reduction with FMA, exact for
// |n| < 2^20, and minimax kernels on [-pi/4, pi/4] (the classic fdlibm k_sin/k_cos
// coefficient sets), ~40 instructions, abs. error <= 1.4 ulp(1) (glibc: 0.5).  Valid
// for |x| < 2^20 * pi/2; the kernel checks the largest phase of a walker once, while
// staging its layers, and sends walkers beyond 1e6 rad (never seen with physical
// inputs) through an out-of-line ocml-sincos body instead.  NaN / Inf propagate.
// ---------------------------------------------------------------------------
constexpr double SINCOS_CW_LIMIT = 1.0e6;

__device__ __forceinline__ void sincos_cw(double x, double &sn, double &cs)
{
    constexpr double INV_PIO2 = 6.36619772367581382433e-01;
    constexpr double PIO2_1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    constexpr double PIO2_2 = 6.07710050630396597660e-11;  // next 33 bits
    constexpr double PIO2_3 = 2.02226624879595063154e-21;  // pi/2 - PIO2_1 - PIO2_2
    const double fn = rint(x * INV_PIO2);
    const int q = (int)fn;
    const double r0 = fma(-fn, PIO2_1, x);       // exact: fn * PIO2_1 has <= 53 bits
    const double r = fma(-fn, PIO2_2, r0);       // one rounding, captured below
    double lo = fma(-fn, PIO2_2, r0 - r);
    lo = fma(-fn, PIO2_3, lo);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    double s0 = fma(r * z, ps, r);               // sin(r)
    double c0 = fma(z, fma(z, pc, -0.5), 1.0);   // cos(r)
    const double s1 = fma(lo, c0, s0);           // sin(r + lo)
    const double c1 = fma(-lo, s0, c0);          // cos(r + lo)
    // quadrant: q&1 swaps, signs from q&2 / (q+1)&2
    const bool sw = q & 1;
    double ss = sw ? c1 : s1;
    double cc = sw ? s1 : c1;
    // (the sign flips as integer work on the high word -- bit 1 of q resp. q + 1 moved to the sign bit and xor-ed in: three
    // VALU instructions each instead of the compare / negate / select four; the same bits, NaN included)
    const unsigned fs = ((unsigned)q << 30) & 0x80000000u;
    const unsigned fc = ((unsigned)(q + 1) << 30) & 0x80000000u;
    sn = __hiloint2double(__double2hiint(ss) ^ (int)fs, __double2loint(ss));
    cs = __hiloint2double(__double2hiint(cc) ^ (int)fc, __double2loint(cc));
}

// Wave-uniform constants read straight from stage_kernel's global image go through the SCALAR data path
// (s_load into SGPRs; a VALU instruction takes one of them as an operand) when the pointer lives in the constant
// address space: no LDS copy, no ds_read per layer and ~38 fewer VGPRs per lane in the chained-phase loop than
// the LDS broadcast -- which is what removed its register-shuffling moves (34 v_mov_b64 per layer at 8 bins).
typedef const double __attribute__((address_space(4))) *KPtr;
__device__ __forceinline__ KPtr as_scalar_ptr(const double *p) { return (KPtr)(uintptr_t)p; }

template <int NCOL>
struct ColState {
    double v[NCOL][4];
};

// one layer applied to NCOL real column vectors:  v <- A v
template <bool FAST>
__device__ __forceinline__ void phase_sincos(double x, double &sn, double &cs)
{
    if (FAST)
        sincos_cw(x, sn, cs);
    else
        sincos(x, &sn, &cs);
}

// one layer applied to NCOL columns held in the layer's eigen-coordinates (a_p, b_p, a_s, b_s):
// rotate the P pair by the layer's P phase and the S pair by its S phase, then change to the
// next layer's coordinates (c[3..10])
template <int NCOL, class CP = const double *>
__device__ __forceinline__ void apply_layer_trig(ColState<NCOL> &s, CP c, double sx,
                                                 double cx, double se, double ce)
{
    const double g0 = c[3], g1 = c[4], g2 = c[5], g3 = c[6], g4 = c[7], g5 = c[8], g6 = c[9], g7 = c[10];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
        const double ap = s.v[j][0], bp = s.v[j][1], as = s.v[j][2], bs = s.v[j][3];
        const double rap = fma(cx, ap, -(sx * bp));
        const double rbp = fma(sx, ap, cx * bp);
        const double ras = fma(ce, as, -(se * bs));
        const double rbs = fma(se, as, ce * bs);
        s.v[j][0] = fma(g1, rbs, g0 * rap);
        s.v[j][3] = fma(g3, rbs, g2 * rap);
        s.v[j][1] = fma(g5, ras, g4 * rbp);
        s.v[j][2] = fma(g7, ras, g6 * rbp);
    }
}

// the same with the walker's gauge applied (stage_interface, unit): c[3] = c[10] = 1 exactly
template <int NCOL, class CP = const double *>
__device__ __forceinline__ void apply_layer_trig_unit(ColState<NCOL> &s, CP c, double sx,
                                                      double cx, double se, double ce)
{
    const double g1 = c[4], g2 = c[5], g3 = c[6], g4 = c[7], g5 = c[8], g6 = c[9];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
        const double ap = s.v[j][0], bp = s.v[j][1], as = s.v[j][2], bs = s.v[j][3];
        // Operation order chosen for the register allocator, not the reader: each rotated value stays live past the
        // fma that adds it to a state slot, so hipcc emits that fma in its three-address form; written the other
        // way round it picks v_fmac and copies the addend into the slot first (4 v_mov_b64 per bin: 6 % of the loop)
        const double u1 = sx * ap, u2 = sx * bp, u3 = se * as, u4 = se * bs;
        const double rap = fma(cx, ap, -u2);
        const double rbp = fma(cx, bp, u1);
        const double ras = fma(ce, as, -u4);
        const double rbs = fma(ce, bs, u3);
        s.v[j][0] = fma(g1, rbs, rap);
        s.v[j][3] = fma(g2, rap, g3 * rbs);
        s.v[j][2] = fma(g6, rbp, ras);
        s.v[j][1] = fma(g5, ras, g4 * rbp);
    }
}

// The constants of one layer that the chained-phase loop uses, as values: the loop fetches the NEXT layer's set while
// it works on the current one (scalar loads have no other latency hiding: a wave that waits for its s_load at the
// top of every layer idles a few hundred cycles per layer).
struct LayerK {
    double xi, eta, h;
    double g1, g2, g3, g4, g5, g6;              // c[4..9]: G in the unit gauge (c[3] = c[10] = 1)
    double px_hi, px_lo, pe_hi, pe_lo;          // c[11..14]
    double Cx, Sx, Ce, Se;                      // c[15..18]
};

template <class CP>
__device__ __forceinline__ LayerK load_layer_k(CP c)
{
    LayerK k;
    k.xi = c[0]; k.eta = c[1]; k.h = c[2];
    k.g1 = c[4]; k.g2 = c[5]; k.g3 = c[6]; k.g4 = c[7]; k.g5 = c[8]; k.g6 = c[9];
    k.px_hi = c[11]; k.px_lo = c[12]; k.pe_hi = c[13]; k.pe_lo = c[14];
    k.Cx = c[15]; k.Sx = c[16]; k.Ce = c[17]; k.Se = c[18];
    return k;
}

template <int NCOL>
__device__ __forceinline__ void apply_layer_trig_unit(ColState<NCOL> &s, const LayerK &k, double sx, double cx, double se,
                                                      double ce)
{
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
        const double ap = s.v[j][0], bp = s.v[j][1], as = s.v[j][2], bs = s.v[j][3];
        // (same operation order as the pointer form above)
        const double u1 = sx * ap, u2 = sx * bp, u3 = se * as, u4 = se * bs;
        const double rap = fma(cx, ap, -u2);
        const double rbp = fma(cx, bp, u1);
        const double ras = fma(ce, as, -u4);
        const double rbs = fma(ce, bs, u3);
        s.v[j][0] = fma(k.g1, rbs, rap);
        s.v[j][3] = fma(k.g2, rap, k.g3 * rbs);
        s.v[j][2] = fma(k.g6, rbp, ras);
        s.v[j][1] = fma(k.g5, ras, k.g4 * rbp);
    }
}

// FAST: the walker's constants carry the unit gauge (the fast paths run only for such walkers)
template <int NCOL, bool FAST, class CP = const double *>
__device__ __forceinline__ void apply_layer(ColState<NCOL> &s, CP c, double omg)
{
    double sx, cx, se, ce;
    // argument formed exactly like the reference: (omega * xi) * z  (forward.f90:397-400)
    phase_sincos<FAST>((omg * c[0]) * c[2], sx, cx);
    phase_sincos<FAST>((omg * c[1]) * c[2], se, ce);
    if (FAST)
        apply_layer_trig_unit<NCOL>(s, c, sx, cx, se, ce);
    else
        apply_layer_trig<NCOL>(s, c, sx, cx, se, ce);
}

// T_rj = sum_k (E^-1 T)(r,k) B_kj  for r = 3 (g[0..3]) or 4 (g[4..7])
template <class CP = const double *>
__device__ __forceinline__ double2 halfspace_row(CP g, const double *v)
{
    return make_double2(fma(g[1], v[3], g[0] * v[0]), fma(g[3], v[2], g[2] * v[1]));
}

template <bool FAST>
__device__ __forceinline__ void cdivq(double2 a1, double2 a2, double2 b, double2 &x1, double2 &x2)
{
    if constexpr (FAST)
        cdiv2_norm(a1, a2, b, x1, x2);
    else
        cdiv2(a1, a2, b, x1, x2);
}

template <int NCOL, bool FAST, class CP = const double *>
__device__ __forceinline__ void finish_bin(const ColState<NCOL> &s, CP tail, double omg,
                                           int ipha, double2 &ur, double2 &uz)
{
    const double2 t31 = halfspace_row(tail, s.v[0]);
    const double2 t41 = halfspace_row(tail + 4, s.v[0]);
    const double2 t32 = halfspace_row(tail, s.v[1]);
    const double2 t42 = halfspace_row(tail + 4, s.v[1]);
    // sl(r,1) = T_r1 ; sl(r,2) = -i T_r2
    const double2 sl31 = t31, sl41 = t41;
    const double2 sl32 = make_double2(t32.y, -t32.x);
    const double2 sl42 = make_double2(t42.y, -t42.x);
    if (NCOL == 2) {
        // free surface (forward.f90:267-275)
        // sl31 sl42 - sl32 sl41 as two four-term chains (1 mul + 3 fma each; the products' roundings differ from
        // csub(cmul, cmul) in the last bit -- the generic path keeps that form)
        const double2 denom = FAST ? make_double2(fma(sl31.x, sl42.x, fma(-sl31.y, sl42.y, fma(-sl32.x, sl41.x, sl32.y * sl41.y))),
                                                  fma(sl31.x, sl42.y, fma(sl31.y, sl42.x, fma(-sl32.x, sl41.y, -(sl32.y * sl41.x)))))
                                   : csub(cmul(sl31, sl42), cmul(sl32, sl41));
        if (ipha >= 0)
            cdivq<FAST>(sl42, cneg(sl41), denom, ur, uz);
        else
            cdivq<FAST>(cneg(sl32), sl31, denom, ur, uz);
    } else {
        // sea floor (forward.f90:276-287).  sl(r,4) = -(i/w) T_r4 and
        // lq21 = -(rho_w w / xi_w) sin: the w cancels in sl(r,4) * lq21.
        const double2 t34 = halfspace_row(tail, s.v[NCOL - 1]);
        const double2 t44 = halfspace_row(tail + 4, s.v[NCOL - 1]);
        double sw, cw;
        phase_sincos<FAST>((omg * tail[8]) * tail[9], sw, cw);
        const double q = tail[10] * sw;                       // (rho_w / xi_w) sin
        const double2 s44l = make_double2(-t44.y * q, t44.x * q); // i T44 q
        const double2 s34l = make_double2(-t34.y * q, t34.x * q);
        const double2 a = cadd(make_double2(sl42.x * cw, sl42.y * cw), s44l);
        const double2 b = cadd(make_double2(sl32.x * cw, sl32.y * cw), s34l);
        // the reference divides uz by d2 = b sl41 - a sl31 = -d1 (exactly): same quotient as -num / d1
        const double2 d1 = csub(cmul(a, sl31), cmul(b, sl41));
        if (ipha >= 0)
            cdivq<FAST>(a, make_double2(-(cw * sl41.x), -(cw * sl41.y)), d1, ur, uz);
        else
            cdivq<FAST>(cneg(b), make_double2(cw * sl31.x, cw * sl31.y), d1, ur, uz);
    }
}

struct SpectraParams {
    DeviceTables t;
    BatchArgs b;
    double2 *spec;
    int nsplit;
    int *slow_list;   // [nslots * nfwd] (walker, trace) pairs deferred to spectra_slow_kernel
    int *slow_count;  // [1] reset by logl_kernel
    WalkerState w;    // stage_kernel's constants (gcoef, gtail, gflag)
    int ablate;       // RFGPU_DIAGNOSTICS builds only (7: the chained path skips boundary condition + deposit)
};

// One wave (64 lanes) per (walker, forward-trace, bin-split).  Lanes own frequency
// bins (coalesced 16-B stores of the spectra); the layer stack of the walker is staged
// in LDS as precomputed coefficients and broadcast to all lanes; the 4x4 chain runs in
// registers.

template <int NCOL, class CP = const double *>
__device__ __forceinline__ void init_cols(ColState<NCOL> &st, CP tail)
{
    const CP t = tail + 11;   // stage_start
#pragma unroll
    for (int j = 0; j < NCOL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st.v[j][r] = 0.0;
    st.v[0][0] = t[0];
    st.v[0][3] = t[1];
    st.v[1][1] = t[2];
    st.v[1][2] = t[3];
    if constexpr (NCOL == 3) {
        st.v[2][0] = t[4];
        st.v[2][3] = t[5];
    }
}

__device__ __forceinline__ void store_bin(double2 *__restrict__ out_r, double2 *__restrict__ out_v, int k, int nh,
                                          double2 ur, double2 uz)
{
    if (k < nh) {
        out_r[k] = make_double2(ur.x, -ur.y);  // freq_r = conjg(ur)   forward.f90:145
        out_v[k] = make_double2(-uz.x, uz.y);  // freq_v = -conjg(uz)  forward.f90:146
    }
}

// where a finished bin goes: to the spectra buffer in HBM (split kernels) ...
struct GlobalSink {
    double2 *__restrict__ out_r;
    double2 *__restrict__ out_v;
    int nh;
    // (the sinks that apply the Gaussian filter hand out its weight ahead of the bin: see LdsSink)
    __device__ __forceinline__ double weight(int) const { return 1.0; }
    // (last argument: the bin's slot within the lane's chain, -1 for a leftover iteration; only RegSink uses it)
    __device__ __forceinline__ void operator()(int k, double2 ur, double2 uz, double, int = 0) const
    {
        store_bin(out_r, out_v, k, nh, ur, uz);
    }
};

// one 64-bin iteration, every phase by a full sincos evaluation
template <int NCOL, bool FAST, class Sink, class CP = const double *>
__device__ __forceinline__ void spectra_iter_direct(const SpectraParams &P, CP coef, CP tail,
                                                    int nl, int ilay0, int ipha, const Sink &sink, int it, int lane)
{
    const int k = it * 64 + lane;
    // forward.f90:245-248: omega = (iomg-1) * domg, DC bin uses the single literal 1.0e-5
    const double omg = k == 0 ? P.t.omg_dc : (double)k * P.t.domg;
    const double wgt = sink.weight(k);
    ColState<NCOL> st;
    init_cols<NCOL>(st, tail);
    for (int l = ilay0; l < nl - 1; ++l) apply_layer<NCOL, FAST>(st, coef + l * NCOEF, omg);
    double2 ur, uz;
    finish_bin<NCOL, FAST>(st, tail, omg, ipha, ur, uz);
    sink(k, ur, uz, wgt, -1);
}

// The iteration that holds the Nyquist bin has one active lane (nfft / 2 is a multiple of 64).  A whole wave used to
// walk the layer stack for it after its own chunk -- ~30 instructions per layer on ONE wave's critical path while
// the block's other waves waited at the barrier (+6 % of the 8-bin chain, +15 % of the 4-bin chain of the 8-wave
// kernels).  Now stage_kernel, which forms the sines and cosines of that bin's phases anyway, runs the bin's chain
// too (stage_nyquist: the same functions, the same arguments as spectra_iter_direct would use) and the kernels only
// deposit the result.
template <int NCOL>
__device__ __forceinline__ void stage_nyquist(const double *nyq, const double *tail, int nl, int ilay0, int ipha,
                                              double omg_nyq, double2 &ur, double2 &uz)
{
    // nyq[l][0..5] = c[4..9] of layer l, [6..9] = c[19..22] (stage_kernel's LDS row); tail: the walker constants
    ColState<NCOL> st;
    init_cols<NCOL>(st, tail);
    for (int l = ilay0; l < nl - 1; ++l) {
        const double *r = nyq + (size_t)l * 10;
        LayerK k;
        k.g1 = r[0]; k.g2 = r[1]; k.g3 = r[2]; k.g4 = r[3]; k.g5 = r[4]; k.g6 = r[5];
        apply_layer_trig_unit<NCOL>(st, k, r[6], r[7], r[8], r[9]);
    }
    finish_bin<NCOL, true>(st, tail, omg_nyq, ipha, ur, uz);
}

template <class Sink, class CP = const double *>
__device__ __forceinline__ void spectra_iter_nyquist(CP tail, const Sink &sink, int it, int lane)
{
    const int k = it * 64 + lane;                // lane 0: the Nyquist bin; the sinks drop the bins beyond it
    const double wgt = sink.weight(k);
    sink(k, make_double2(tail[18], tail[19]), make_double2(tail[20], tail[21]), wgt, -1);
}

// eps = arg - k * phi, phi = (hi, lo): the rounding perturbation of the reference's argument (omega*xi)*z
// relative to the exact multiple k*phi.  k*phi_hi - arg is formed by ONE fma and is exact: the 65-bit product and
// the 53-bit argument agree to a few ulps, so their difference has ~15 significant bits; the second fma adds
// k*phi_lo with a single rounding (2^-53 of eps).  Two instructions.
__device__ __forceinline__ double phase_eps(double arg, double kd, double phi_hi, double phi_lo)
{
    return -fma(kd, phi_lo, fma(kd, phi_hi, -arg));
}

// ---------------------------------------------------------------------------
// Anchor table of fused8_kernel.  The chain of a lane starts at bin k0 = 64 BK ch + lane (ch: the wave's chunk),
// so the exact-angle pair it starts from factors into a part every wave of the block shares and a wave-uniform part:
//     E(k0 phi) = E(64 BK ch phi) * E(lane phi),        E(x) = (cos x, sin x).
// The block computes both factors ONCE per (layer, phase) -- 64 + nchunk evaluations of sincos_cw instead of one per
// lane and wave -- into LDS, in the space of the FFT array, which nothing else uses until the bins are deposited
// (a barrier separates the two uses).  Row (l - ilay0, phase) holds ANCHOR_ROW double2 entries (cos, sin):
// [0, 64) the lane factors, [64, 64 + nchunk) the chunk factors (entry 64 and entry 0 are exactly (1, 0)).
// Each entry is the sine and cosine of the real number j * phi, phi = (hi, lo): the rounded product's sincos,
// corrected to first order by the product's rounding error and j * lo (|err| < 1e-13).
// ---------------------------------------------------------------------------
constexpr int ANCHOR_ROW = 72;

__device__ __forceinline__ double2 exact_angle(double mult, double phi_hi, double phi_lo)
{
    const double x = mult * phi_hi;
    const double err = fma(mult, phi_lo, fma(mult, phi_hi, -x));   // (mult * phi) - x: the inner fma is exact
    double s, c;
    sincos_cw(x, s, c);
    return make_double2(fma(-s, err, c), fma(c, err, s));
}

__device__ __forceinline__ double readlane_f64(double v, int src)   // src: wave-uniform
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// gc: stage_kernel's image of this (item, trace); nsolid <= 64.  Lane i of every wave fetches the four phase doubles
// of solid layer i -- ONE round trip to memory for the whole table -- and the passes below take them from there.
// Lane factors: a whole-wave pass per layer (both phases: two independent evaluations); wave 0 takes half a share
// (its chunk holds the DC bin, whose first-bin evaluation is longer: spectra_chunk_chain).  Chunk factors of all
// layers: packed 64 to a pass.
template <int THREADS>
__device__ __forceinline__ void build_anchor_table(double2 *tab, const double *__restrict__ gc, int nsolid, int ilay0,
                                                   int nchunk, int chunk_bins, int tid)
{
    constexpr int NW = THREADS / 64;
    const int wave = tid >> 6, lane = tid & 63;
    double xh = 0.0, xl = 0.0, eh = 0.0, el = 0.0;
    if (lane < nsolid) {
        const double *q = gc + (size_t)(ilay0 + lane) * NCOEF + 11;
        xh = q[0]; xl = q[1]; eh = q[2]; el = q[3];
    }
    const double m = (double)lane;
    for (int li = 0; li < nsolid; ++li) {
        const int c = li % (2 * NW - 1);
        const int owner = c < NW - 1 ? c + 1 : (c == NW - 1 ? 0 : c - NW + 1);
        if (owner != wave) continue;
        double2 *row = tab + 2 * li * ANCHOR_ROW + lane;
        row[0] = exact_angle(m, readlane_f64(xh, li), readlane_f64(xl, li));
        row[ANCHOR_ROW] = exact_angle(m, readlane_f64(eh, li), readlane_f64(el, li));
    }
    const int per = 2 * nchunk, nfac = nsolid * per;
    for (int p = wave; p * 64 < nfac; p += NW) {
        const int q = min(p * 64 + lane, nfac - 1);
        const int li = q / per, rem = q - li * per;
        const int phs = rem >= nchunk ? 1 : 0, ch = rem - phs * nchunk;
        const double axh = __shfl(xh, li, 64), axl = __shfl(xl, li, 64), aeh = __shfl(eh, li, 64), ael = __shfl(el, li, 64);
        const double2 e = exact_angle((double)(ch * chunk_bins), phs ? aeh : axh, phs ? ael : axl);
        if (p * 64 + lane < nfac) tab[(2 * li + phs) * ANCHOR_ROW + 64 + ch] = e;
    }
}

// (cos, sin)(x) for |x| < 2^-8 by its series (error < 5e-18): the DC bin's phase is omega_dc xi h with the
// single-precision literal omega_dc = 1e-5 (forward.f90:247), far too large for a first-order correction of the
// chain's angle 0 and far too small to be worth a range reduction.  stage_kernel sends walkers with a larger DC
// phase (a layer thousands of kilometres thick) to the generic path.
constexpr double DC_PHASE_LIMIT = 0x1p-8;
__device__ __forceinline__ void sincos_small(double x, double &sn, double &cs)
{
    const double z = x * x;
    cs = fma(z, fma(z, 1.0 / 24.0, -0.5), 1.0);
    sn = fma(x * z, fma(z, 1.0 / 120.0, -1.0 / 6.0), x);
}

// BK consecutive 64-bin iterations per lane (bins k0, k0+64, ...): "chained phases".
// For each layer only the first bin pays a full sincos; the exact-angle pair
// (cos, sin)(k phi) then advances by the wave-uniform rotation (cos, sin)(64 phi) of the
// layer, and each bin's actual phase -- the reference's rounded double (omega*xi)*z, whose
// rounding is part of the reference result -- is recovered to first order from
// eps = arg - k phi (|eps| < 1e-9, second order < 1e-18).  ~13 instructions per extra
// sincos instead of ~45.  Chain length <= BK-1 rotations (error growth ~1 ulp per step).
// TABLE (fused8_kernel): the chain's starting pair comes from the block's anchor table (one complex product) and
// the first bin is corrected like every other; its only special case is the DC bin (lane 0 of chunk 0).  A barrier
// separates the layer loop (table reads) from the deposit of the bins, which overwrites the table: every wave of
// the block runs exactly one chunk in this mode.
template <int BK, int NCOL, class Sink, class CP = const double *, bool TIGHT = false, bool TABLE = false>
__device__ __forceinline__ void spectra_chunk_chain(const SpectraParams &P, CP coef, CP tail,
                                                    int nl, int ilay0, int ipha, const Sink &sink, int it0, int lane,
                                                    const double2 *tab = nullptr, double2 *keep_ur = nullptr,
                                                    double2 *keep_uz = nullptr)
{
    // Register budget (two waves per SIMD: 256 VGPRs; TIGHT: the 128 of fused8_kernel).  Short 2-column chains hold
    // k and omega of every bin; the 8-bin land kernel and the 3-column ocean kernels hold the omegas and rebuild k
    // (one exact addition per bin and layer); the TIGHT chains (and the 8-bin ocean chain nothing selects) keep only
    // the first bin's index and rebuild both where they are used (the values are the same doubles).
    constexpr bool LEAN = TIGHT || (NCOL != 2 && BK >= 8);
    constexpr bool KEEP_KD = !LEAN && BK < 8 && NCOL == 2;
    ColState<NCOL> st[BK];
    double omg[LEAN ? 1 : BK], kd[KEEP_KD ? BK : 1];
    const int k0 = it0 * 64 + lane;
#pragma unroll
    for (int m = 0; m < BK; ++m) {
        if (!LEAN) {
            const double kdm = (double)(k0 + 64 * m);
            if (KEEP_KD) kd[m] = kdm;
            omg[m] = (k0 + 64 * m) == 0 ? P.t.omg_dc : kdm * P.t.domg;
        }
        init_cols<NCOL>(st[m], tail);
    }
    const bool dc = k0 == 0;
    int kk = k0;
    const double2 *trow = TABLE ? tab + lane : nullptr;                 // lane factor of (layer, phase) row r: trow[r * ANCHOR_ROW]
    const double2 *tfac = TABLE ? tab + 64 + it0 / BK : nullptr;        // chunk factor of row r: tfac[r * ANCHOR_ROW]
    static_assert(std::is_same<CP, KPtr>::value, "the chained-phase path reads stage_kernel's image through SGPRs");

    // one layer applied to the BK bins of the lane, constants c
    auto layer_step = [&](const LayerK &c, int l) {
        const double xi = c.xi, eta = c.eta, h = c.h;
        // k and omega of the first bin: long chains rebuild them from the bin index in every layer (kk goes
        // through the loop's asm, which keeps the compiler from hoisting them back into registers that it would
        // then have to spill: 4 instructions per layer against 8 scratch reloads)
        const double kd0l = (double)kk;
        const double omg0l = LEAN ? (kk == 0 ? P.t.omg_dc : kd0l * P.t.domg) : omg[0];
        // first bin: the reference's argument
        const double ax0 = (omg0l * xi) * h, ae0 = (omg0l * eta) * h;
        const double ex0 = phase_eps(ax0, kd0l, c.px_hi, c.px_lo);
        const double ee0 = phase_eps(ae0, kd0l, c.pe_hi, c.pe_lo);
        double cEx, sEx, cEe, sEe;
        if constexpr (TABLE) {
            // exact-angle start of the chain: chunk factor * lane factor
            const int r = 2 * (l - ilay0) * ANCHOR_ROW;
            const double2 bx = trow[r], be = trow[r + ANCHOR_ROW];
            const double2 fx = tfac[r], fe = tfac[r + ANCHOR_ROW];
            cEx = fma(fx.x, bx.x, -(fx.y * bx.y));
            sEx = fma(fx.y, bx.x, fx.x * bx.y);
            cEe = fma(fe.x, be.x, -(fe.y * be.y));
            sEe = fma(fe.y, be.x, fe.x * be.y);
            double sx = fma(cEx, ex0, sEx), cx = fma(-sEx, ex0, cEx), se = fma(cEe, ee0, sEe), ce = fma(-sEe, ee0, cEe);
            if (it0 == 0) {
                // (wave-uniform) the DC bin: angle 0 + the phase of omega_dc, by series
                double s1, c1, s2, c2;
                sincos_small(ax0, s1, c1);
                sincos_small(ae0, s2, c2);
                sx = dc ? s1 : sx;
                cx = dc ? c1 : cx;
                se = dc ? s2 : se;
                ce = dc ? c2 : ce;
            }
            apply_layer_trig_unit<NCOL>(st[0], c, sx, cx, se, ce);
        } else {
            // direct evaluation; then the exact-angle start of the chain: remove the first bin's own
            // perturbation.  The DC bin's omega is the literal 1e-5 (not 0 * domg): its chain starts from angle 0.
            double sx, cx, se, ce;
            sincos_cw(ax0, sx, cx);
            sincos_cw(ae0, se, ce);
            apply_layer_trig_unit<NCOL>(st[0], c, sx, cx, se, ce);
            // (round 6: testing it0 == 0 first -- wave-uniform, only that chunk holds the DC bin -- so that the other
            // waves skip these eight selects was measured: C4 -0.5 %, but C2 +5.5 % and C3 +1 % (the eight-wave kernels'
            // register assignment shifts) and 29 instead of 19 spills in the ocean chain; profiles/EXPERIMENTS.md)
            cEx = dc ? 1.0 : fma(sx, ex0, cx);
            sEx = dc ? 0.0 : fma(-cx, ex0, sx);
            cEe = dc ? 1.0 : fma(se, ee0, ce);
            sEe = dc ? 0.0 : fma(-ce, ee0, se);
        }
#pragma unroll
        for (int m = 1; m < BK; ++m) {
            const double tx = cEx, te = cEe;
            cEx = fma(tx, c.Cx, -(sEx * c.Sx));
            sEx = fma(sEx, c.Cx, tx * c.Sx);
            cEe = fma(te, c.Ce, -(sEe * c.Se));
            sEe = fma(sEe, c.Ce, te * c.Se);
            const double kdm = KEEP_KD ? kd[m] : kd0l + (double)(64 * m);    // exact: small integers
            const double omgm = LEAN ? kdm * P.t.domg : omg[m];
            const double ex = phase_eps((omgm * xi) * h, kdm, c.px_hi, c.px_lo);
            const double ee = phase_eps((omgm * eta) * h, kdm, c.pe_hi, c.pe_lo);
            apply_layer_trig_unit<NCOL>(st[m], c, fma(cEx, ex, sEx), fma(-sEx, ex, cEx), fma(cEe, ee, sEe),
                                        fma(-sEe, ee, cEe));
        }
    };
    // (wave-uniform address of a layer's record; under register pressure the compiler may hold the base in VGPRs,
    // and it does not legalise an inline asm's "s" operand by itself)
    auto record_of = [&](int l) -> uint64_t {
        const uint64_t an = (uint64_t)(uintptr_t)(coef + l * NCOEF);
        return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(an >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)an);
    };
    // Scalar-cache prefetch of the next layer's record (three 64-byte lines): by the next iteration it sits in the
    // scalar cache and that iteration's s_loads, which the wave waits for before it can do anything, are hits.
    // (Fetching the next layer's constants themselves into a second register set a layer ahead -- 17 s_loads by
    // inline asm, loop unrolled by two -- was measured: 13 % slower at C4; the scalar moves and loads it adds
    // compete with the loop's own issue.)
    // Inline asm because the compiler sinks an ordinary load down to its use; tied to the first bin's index (a VGPR
    // operand the iteration needs at once) so that it stays at the top.  The three destination registers are
    // loop-carried ("+s"): reserved for the whole loop, never read; the wait after the loop retires the last loads
    // before the registers are handed back.
    unsigned touch0 = 0, touch1 = 0, touch2 = 0;
#pragma unroll 1
    for (int l = ilay0; l < nl - 1; ++l) {
        LayerK c = load_layer_k(coef + l * NCOEF);
        // (every constant through one empty asm: the compiler then issues all the loads together and waits once;
        // left alone it sometimes splits them around the first uses and the wave waits twice per layer, +3 %)
        asm volatile("" : "+s"(c.xi), "+s"(c.eta), "+s"(c.h), "+s"(c.g1), "+s"(c.g2), "+s"(c.g3), "+s"(c.g4), "+s"(c.g5),
                     "+s"(c.g6), "+s"(c.px_hi), "+s"(c.px_lo), "+s"(c.pe_hi), "+s"(c.pe_lo), "+s"(c.Cx), "+s"(c.Sx),
                     "+s"(c.Ce), "+s"(c.Se));
        const uint64_t cn = record_of(l + 1 < nl - 1 ? l + 1 : l);
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80"
                     : "+s"(touch0), "+s"(touch1), "+s"(touch2), "+v"(kk)
                     : "s"(cn));
        layer_step(c, l);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(touch0), "+s"(touch1), "+s"(touch2));
    if constexpr (TABLE) __syncthreads();   // every wave is done with the table: the deposits below overwrite it
    if (RFGPU_ABLATE_IS(7)) {
        // timing split (diagnostics builds): the layer loop alone (one store keeps the chain alive)
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < BK; ++m)
#pragma unroll
            for (int j = 0; j < NCOL; ++j) acc += st[m].v[j][0] + st[m].v[j][1] + st[m].v[j][2] + st[m].v[j][3];
        if (acc == 1.2345e300) sink(k0, make_double2(acc, 0.0), make_double2(0.0, acc), 1.0, 0);
        return;
    }
    // the bins' filter weights: every load is in flight before the first boundary condition is evaluated (the
    // chain's registers are free by now); loaded where they are used, each bin waited for its own
    // (the 8-bin ocean chain holds 192 state registers at this point: its weights are fetched one bin ahead instead)
    // and its bin indices are rebuilt from the copy that went through the loop's asm, which nothing derived from can
    // be hoisted above the loop into registers that would be spilled there)
    constexpr bool ROLL = NCOL != 2 && BK >= 8;
    const int kf = ROLL ? kk : k0;
    double wgt[BK];
#pragma unroll
    for (int m = 0; m < (ROLL ? 1 : BK); ++m) wgt[m] = sink.weight(kf + 64 * m);
#pragma unroll
    for (int m = 0; m < BK; ++m) {
        const int km = kf + 64 * m;
        const double omgm = km == 0 ? P.t.omg_dc : (double)km * P.t.domg;
        if (ROLL && m + 1 < BK) wgt[m + 1] = sink.weight(km + 64);
        double2 ur, uz;
        finish_bin<NCOL, true>(st[m], tail, omgm, ipha, ur, uz);
        if (keep_ur) {
            // fusedc_kernel: the lane keeps its bins (the caller's register arrays; constant indices after unrolling)
            keep_ur[m] = ur;
            keep_uz[m] = uz;
        } else {
            sink(km, ur, uz, wgt[m], m);
        }
    }
}

// the bins of one (walker, forward-trace) assigned to `split` of P.nsplit:
// full chunks of BK iterations go through the chained-phase path, the remaining
// iterations (and everything when BK == 0) through the direct path.
template <int BK, int NCOL, bool FAST, class Sink, class CP = const double *, bool TIGHT = false, bool TABLE = false>
__device__ __forceinline__ void spectra_body(const SpectraParams &P, CP coef, CP tail,
                                             int nl, int ilay0, int ipha, const Sink &sink, int split, int lane,
                                             const double2 *tab = nullptr, double2 *keep_ur = nullptr,
                                             double2 *keep_uz = nullptr)
{
    const int niter = (P.t.nh + 63) / 64;
    int it_direct0 = 0;
    if constexpr (BK > 1 && FAST) {
        const int nchunk = niter / BK;
        if constexpr (TABLE) {
            // (the caller checked nchunk == P.nsplit: one chunk per wave, see spectra_chunk_chain)
            spectra_chunk_chain<BK, NCOL, Sink, CP, TIGHT, true>(P, coef, tail, nl, ilay0, ipha, sink, split * BK, lane, tab,
                                                                 keep_ur, keep_uz);
        } else {
            // (keep_ur: the caller runs exactly one chunk per wave, nchunk == P.nsplit)
            for (int ch = split; ch < nchunk; ch += P.nsplit)
                spectra_chunk_chain<(BK > 1 ? BK : 2), NCOL, Sink, CP, TIGHT>(P, coef, tail, nl, ilay0, ipha, sink, ch * BK, lane,
                                                                              nullptr, keep_ur, keep_uz);
        }
        it_direct0 = nchunk * BK;
    }
    // leftover iterations: spread from the last split downwards (the chunk loop loads
    // the low splits first)
    for (int it = it_direct0 + (P.nsplit - 1 - split); it < niter; it += P.nsplit) {
        if (FAST && it > 0 && it * 128 == P.t.nfft)
            spectra_iter_nyquist(tail, sink, it, lane);
        else
            spectra_iter_direct<NCOL, FAST>(P, coef, tail, nl, ilay0, ipha, sink, it, lane);
    }
}

// Copies stage_kernel's constants of (batch item ib, forward-trace f) into the block's LDS image (all threads of
// the block; coalesced 16-byte loads, ~6 KB at 30 layers).  Returns true (block-uniform) when the walker must take
// the generic path (a phase beyond the Cody-Waite range of sincos_cw, or no unit gauge).  tail[17] = the
// direct-arrival time of the forward trace.
__device__ __forceinline__ bool load_staged(const WalkerState &w, const BatchArgs &b, int nfwd, int ib, int f, double *coef,
                                            double *tail, int &nl, int &ilay0, bool &sea)
{
    const int bfi = ib * nfwd + f, pad = b.nlay_pad;
    nl = b.nlay[ib];
    const int fl = w.gflag[bfi];
    sea = fl & 1;                     // beta(1) < 0  (forward.f90:229)
    ilay0 = sea ? 1 : 0;
    const double2 *__restrict__ gc = reinterpret_cast<const double2 *>(w.gcoef + (size_t)bfi * pad * NCOEF);
    double2 *lc = reinterpret_cast<double2 *>(coef);
    for (int i = ilay0 * (NCOEF / 2) + (int)threadIdx.x; i < (nl - 1) * (NCOEF / 2); i += blockDim.x) lc[i] = gc[i];
    const double *__restrict__ gt = w.gtail + (size_t)bfi * GTAIL;
    if (threadIdx.x < GTAIL) tail[threadIdx.x] = gt[threadIdx.x];
    __syncthreads();
    return (fl & 2) != 0;
}

// WPB waves per block share one staged layer stack; wave w of block b works on split
// (b % nblk) * WPB + w of the walker.  NCOL = 2 is the land kernel, NCOL = 3 the ocean
// kernel (one more column of the product is propagated); the host launches the one that
// matches params' sdep, and a walker whose own beta(1) < 0 test (forward.f90:229)
// disagrees with it is deferred like an out-of-range one.
template <int BK, int NCOL>
__global__ __launch_bounds__(256) void spectra_kernel(SpectraParams P)
{
    extern __shared__ double lds[];
    const int wpb = blockDim.x >> 6;
    const int nblk = (P.nsplit + wpb - 1) / wpb;       // blocks per (walker, forward-trace)
    const int bf = blockIdx.x / nblk;
    const int split = (blockIdx.x % nblk) * wpb + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int f = bf % P.t.nfwd;
    const int ib = P.b.order ? P.b.order[bf / P.t.nfwd] : bf / P.t.nfwd;
    if (P.w.item_state[ib] != 1) return;      // (stage_kernel's verdict: evaluate / sigma-only / skipped / refused)

    // stage_kernel's constants: the fast paths read them through the scalar data path; only a walker on the generic
    // path (rare) copies the image into LDS
    const int bfi = ib * P.t.nfwd + f;   // (walker, forward-trace) index in batch order
    const int nl = P.b.nlay[ib];
    const int stage_flags = P.w.gflag[bfi];
    const bool sea = stage_flags & 1;    // beta(1) < 0  (forward.f90:229)
    const int ilay0 = sea ? 1 : 0;
    const int ipha = P.t.ipha[f];
    double2 *out_r = P.spec + ((size_t)(ib * P.t.nfwd + f) * 2) * P.t.nh;
    const GlobalSink sink{out_r, out_r + P.t.nh, P.t.nh};
    if ((stage_flags & 2) != 0 || sea != (NCOL == 3)) {
        // rare: out-of-range phases, no unit gauge, or a layer stack of the other kind (land / ocean)
        if (BK > 1) {
            // the chained-phase kernels have the registers to spare: generic path in place
            double *coef = lds;
            double *tail = lds + (size_t)P.b.nlay_pad * NCOEF;
            int nl2, il2;
            bool sea2;
            (void)load_staged(P.w, P.b, P.t.nfwd, ib, f, coef, tail, nl2, il2, sea2);
            if (split >= P.nsplit) return;
            if (sea)
                spectra_body<0, 3, false>(P, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, split, lane);
            else
                spectra_body<0, 2, false>(P, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, split, lane);
        } else if (blockIdx.x % nblk == 0 && threadIdx.x == 0) {
            // the lean direct kernel (4 waves/SIMD) defers to spectra_slow_kernel via the list
            P.slow_list[atomicAdd(P.slow_count, 1)] = bfi;
        }
        return;
    }
    if (split >= P.nsplit) return;
    const KPtr gcoef = as_scalar_ptr(P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF);
    const KPtr gtail = as_scalar_ptr(P.w.gtail + (size_t)bfi * GTAIL);
    spectra_body<BK, NCOL, true>(P, gcoef, gtail, nl, ilay0, ipha, sink, split, lane);
}

// walkers whose phases exceed the Cody-Waite range (|x| >= 1e6 rad): same body with ocml's
// generic sincos, in its own kernel so its registers do not burden the fast path.  Fixed
// small grid striding over the (normally empty) list.
__global__ __launch_bounds__(64) void spectra_slow_kernel(SpectraParams P)
{
    extern __shared__ double lds[];
    const int count = *P.slow_count;
    for (int e = blockIdx.x; e < count * P.nsplit; e += gridDim.x) {
        const int bf = P.slow_list[e / P.nsplit];
        const int split = e % P.nsplit;
        const int f = bf % P.t.nfwd;
        const int ib = bf / P.t.nfwd;
        double *coef = lds;
        double *tail = lds + (size_t)P.b.nlay_pad * NCOEF;
        int nl, ilay0;
        bool sea;
        __syncthreads();
        load_staged(P.w, P.b, P.t.nfwd, ib, f, coef, tail, nl, ilay0, sea);
        const int ipha = P.t.ipha[f];
        double2 *out_r = P.spec + ((size_t)(ib * P.t.nfwd + f) * 2) * P.t.nh;
        const GlobalSink sink{out_r, out_r + P.t.nh, P.t.nh};
        if (sea)
            spectra_body<0, 3, false>(P, coef, tail, nl, ilay0, ipha, sink, split, threadIdx.x);
        else
            spectra_body<0, 2, false>(P, coef, tail, nl, ilay0, ipha, sink, split, threadIdx.x);
    }
}

size_t spectra_lds_bytes(int nlay_pad) { return sizeof(double) * ((size_t)nlay_pad * (NCOEF + 1) + 24); }

// ---------------------------------------------------------------------------
// K0  stage_kernel: the per-(walker, forward-trace) constants of K1, computed ONCE per batch item by a wide,
// shallow launch in front of the spectra / fused kernel -- not by every block that works on the item, on its
// critical path (a dependent chain of divisions, square roots and three sincos per layer: ~4 us per block,
// 4.5 % of a C4 launch).  A group of G lanes (16, 32 or 64: the smallest that holds the context's layers) per
// (item, forward-trace), one LAYER per lane; the four independent parts of a layer (stage_interface),
//   part 0: the (a_p, b_s) block of G, c[3..6]    (S slownesses)     part 2: xi, h and the P phase constants
//   part 1: the (b_p, a_s) block of G, c[7..10]   (P slownesses)     part 3: eta and the S phase constants
// run one after the other, each by every lane at once -- no lane waits while another part's code runs, and the
// 64 / G groups of a wave execute the same instructions on different items.  (One lane per (layer, part) pair with
// a wave per item, as before: the wave ran the interface code and the phase code one after the other with half
// its lanes idle each time, twice for more than 16 layers -- 2.3x the instructions per item at C4, 4x at C2.)
// Every consumer copies the same image, so the fused and the split launch plans -- and a chain evaluated alone
// or in a batch -- see identical constants.
// Output per bf = item * nfwd + f (global; the LDS image described above K1):
//   gcoef[bf][nlay_pad][NCOEF]   layers ilay0 .. nl-2
//   gtail[bf][GTAIL]             tail[0..16], [17] direct-arrival time (forward.f90:474-519)
//   gflag[bf]                    bit 0: sea (beta(1) < 0), bit 1: generic path (a phase beyond the Cody-Waite
//                                range, a DC phase beyond its series, or no unit gauge: walker_gauge)
// ---------------------------------------------------------------------------
struct StageParams {
    DeviceTables t;
    BatchArgs b;
    double *gcoef, *gtail;
    int *gflag;
    int spread;   // G = 16 only: the four 16-lane groups of a wave share ONE item, one part each (small batches)
    int *item_state;   // [nb] out: what the following kernels do with the item (item_not_evaluated)
    int *err;          // [4] the context's error word (device-mapped host memory): {reason, batch item, offending value, -}
    int nslots;
};

// LDS of stage_kernel per (item, forward-trace) group: the direct-arrival terms [pad], the constants the Nyquist
// bin's chain reads [pad][STAGE_NYQ] (per layer: c[4..9] = G in the unit gauge, c[19..22] = that bin's sines and
// cosines) and the walker constants tail[0..GTAIL) before they go to the global image in one coalesced write
constexpr int STAGE_NYQ = 10;
__host__ __device__ inline size_t stage_row_doubles(int pad) { return (size_t)pad * (1 + STAGE_NYQ) + GTAIL; }

template <int G>
__global__ __launch_bounds__(256) void stage_kernel(StageParams S)
{
    extern __shared__ double lds[];                 // [waves][64 / G] rows of stage_row_doubles(nlay_pad), one per group
    constexpr int NG = 64 / G;                      // groups per wave
    const int pad = S.b.nlay_pad;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / G, sl = lane % G;
    // Small batches (spread, G = 16): latency counts, not throughput -- the wave's four groups take the four parts
    // of ONE item side by side (q = the group's part; the interface code runs once, the phase code once) instead
    // of four items with the parts one after the other.  The arithmetic per (layer, part) is the same code either
    // way: `part` is a run-time value in both modes, so the results are bit-identical.
    const int q = (G == 16 && S.spread) ? grp : -1;
    // position in DISPATCH order (deepest walkers first, when the batch has one): the groups of a wave then hold
    // walkers of similar depth, so a wave of shallow walkers makes one pass over its lanes where a deep one makes two
    const int pos = q >= 0 ? blockIdx.x * (blockDim.x >> 6) + wave : (blockIdx.x * (blockDim.x >> 6) + wave) * NG + grp;
    const bool live = pos < S.b.nb * S.t.nfwd;
    const int ib = live ? (S.b.order ? S.b.order[pos / S.t.nfwd] : pos / S.t.nfwd) : 0, f = live ? pos % S.t.nfwd : 0;
    const int bf = ib * S.t.nfwd + f;                  // index of the (item, forward-trace) images, batch order
    // The one place every item of a batch passes through: what the kernels that follow are to do with it, and the
    // check of what the caller handed over.  The *_device entry points cannot look at their arrays on the host, and a
    // layer count beyond nlay_pad or a walker id beyond the context's slots would index the constants' image, LDS rows
    // and the trace array (8.6 GB at C5) out of bounds: such an item is REFUSED -- state -2, logL = NaN, nothing of it
    // evaluated or written -- and the context's error word says which (rf_last_error at the next call that checks it).
    int state = live ? (S.b.fwd_flag ? S.b.fwd_flag[ib] : 1) : -1;
    if (live) {
        const int wid = S.b.walker_ids[ib], nli = S.b.nlay[ib];
        int why = 0, what = 0;
        if (state > 1) { why = 3; what = state; }
        else if (wid < 0 || wid >= S.nslots) { why = 2; what = wid; }
        else if (state == 1 && (nli < 2 || nli > pad)) { why = 1; what = nli; }
        if (why) {
            state = -2;
            if (f == 0 && sl == 0 && q <= 0 && S.err[0] == 0) {   // (the first report wins; a race between two is harmless)
                S.err[1] = ib;
                S.err[2] = what;
                S.err[0] = why;
            }
        } else if (state < 0) {
            state = -1;
        }
        if (f == 0 && sl == 0 && q <= 0) S.item_state[ib] = state;
    }
    const bool run = live && state == 1;
    const unsigned long long group_mask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1)) << (grp * G);
    double *terms = lds + (size_t)(q >= 0 ? wave : wave * NG + grp) * stage_row_doubles(pad);
    double *nyq = terms + pad;                      // [pad][STAGE_NYQ]
    double *tl = nyq + (size_t)pad * STAGE_NYQ;     // [GTAIL] the walker constants, assembled here
    // The lanes store their per-layer pieces straight into the global image.  (Assembling that image in LDS and
    // writing whole rows was measured: no faster -- the kernel is bound by its divisions and square roots.)
    bool big = false;
    int nl = 2;
    bool sea = false;
    if (run) {
        const double *L = S.b.layers + (size_t)ib * 4 * pad;
        nl = S.b.nlay[ib];
        const double p = S.t.rayps[f];
        sea = L[pad] < 0.0;                         // beta(1) < 0  (forward.f90:229)
        const int ilay0 = sea ? 1 : 0;
        const bool solid = nl - 1 > ilay0;          // at least one solid layer above the half-space
        const double omg_max = (double)(S.t.nh - 1) * S.t.domg;
        const double omg_nyq = (double)(S.t.nfft / 2) * S.t.domg;
        double *coef = S.gcoef + (size_t)bf * pad * NCOEF;
        bool unit;
        const double gauge = walker_gauge<G>(L, pad, nl, ilay0, p, sl, group_mask, unit);
        big = !unit;
        // the walker's nl - 1 layers above the half-space, one per lane
        for (int l = sl; l < nl - 1; l += G) {
            // this layer and the one below it
            const double a0 = L[l], b0 = L[pad + l], r0 = L[2 * pad + l], h0 = L[3 * pad + l];
            const double a1 = L[l + 1], b1 = L[pad + l + 1], r1 = L[2 * pad + l + 1];
            double *c = coef + (size_t)l * NCOEF;
            double *ny = nyq + (size_t)l * STAGE_NYQ;
            if (l >= ilay0) {
                const int part_lo = q < 0 ? 0 : (q < 2 ? q : 2), part_hi = q < 0 ? 2 : (q < 2 ? q + 1 : 2);
                const int sph_lo = q < 0 ? 0 : (q >= 2 ? q - 2 : 2), sph_hi = q < 0 ? 2 : (q >= 2 ? q - 1 : 2);
#pragma unroll 1
                for (int part = part_lo; part < part_hi; ++part) {
                    // part 0: beta (eta), part 1: alpha (xi)
                    const LayerHalf u = layer_half(part ? a0 : b0, b0, r0, p);
                    double g4[4];
                    if (l + 1 < nl - 1) {
                        const LayerHalf w = layer_half(part ? a1 : b1, b1, r1, p);
                        stage_interface(g4, part, u, &w, unit);
                    } else {
                        stage_interface(g4, part, u, nullptr, unit);
                    }
                    c[3 + 4 * part] = g4[0]; c[4 + 4 * part] = g4[1]; c[5 + 4 * part] = g4[2]; c[6 + 4 * part] = g4[3];
                    // (the unit-gauge chain reads c[4..9]: part 0 supplies c[4..6], part 1 c[7..9])
                    ny[3 * part] = g4[1 - part]; ny[3 * part + 1] = g4[2 - part]; ny[3 * part + 2] = g4[3 - part];
                }
#pragma unroll 1
                for (int sph = sph_lo; sph < sph_hi; ++sph) {
                    // sph 0: the P phase (part 2), 1: the S phase (part 3)
                    const double slow = vertical_slowness(sph ? b0 : a0, p);
                    // phases of the Nyquist bin, argument formed like the reference (forward.f90:397-400)
                    double sn, cn;
                    sincos_cw((omg_nyq * slow) * h0, sn, cn);
                    c[sph] = slow;
                    c[2 + 21 * sph] = sph ? 0.0 : h0;      // c[2] = h (part 2), c[23] = pad (part 3)
                    stage_phase(c + 11 + 2 * sph, c + 15 + 2 * sph, S.t.domg, slow, h0);
                    c[19 + 2 * sph] = sn;
                    c[20 + 2 * sph] = cn;
                    ny[6 + 2 * sph] = sn;
                    ny[7 + 2 * sph] = cn;
                    big |= fabs(omg_max * slow * h0) >= SINCOS_CW_LIMIT || !(fabs(S.t.omg_dc * slow * h0) < DC_PHASE_LIMIT);
                }
            }
            // walker constants, by the lanes that already hold the layers involved (spread: of the lighter groups)
            if (l == nl - 2 && (q < 0 || q == 2)) {
                // half-space = layer l + 1; the last solid layer (if any) = layer l
                LayerBasis last;
                if (solid) last = layer_basis(a0, b0, r0, p);
                stage_halfspace(tl, a1, b1, r1, p, solid ? &last : nullptr, gauge);
            }
            if (l == (solid ? ilay0 : 0) && (q < 0 || q == 3)) {
                LayerBasis top;
                if (solid) top = layer_basis(a0, b0, r0, p);
                stage_start(tl + 11, solid ? &top : nullptr);
            }
            if (l == 0 && sea && (q < 0 || q == 3)) {
                const double xiw = vertical_slowness(a0, p);   // forward.f90:431
                tl[8] = xiw;
                tl[9] = h0;
                tl[10] = r0 / xiw;
                big |= fabs(omg_max * xiw * h0) >= SINCOS_CW_LIMIT;
            }
        }
        // direct_arrival (forward.f90:474-519): the independent per-layer terms by separate lanes, summed
        // strictly in layer order below (it feeds nint(): bit-exact bookkeeping)
        const double *vel = (S.t.ipha[f] == 1) ? L : L + pad;   // alpha for P, beta for S (:157,161)
        const int i0 = S.t.sdep > 0.0 ? 1 : 0;                  // keyed on sdep (:484)
        if (q <= 0)
            for (int i = i0 + sl; i < nl - 1; i += G) terms[i - i0] = arrival_term(L[3 * pad + i], vel[i], p);
    }
    const bool any_big = (__ballot(big) & (q >= 0 ? ~0ull : group_mask)) != 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the row is complete: every lane of the group may read it
    // The Nyquist bin, when it sits alone in its 64-bin iteration (nfft a multiple of 128): its chain over the walker's
    // layers, here, from the LDS row the group's lanes have just filled -- every lane runs it (same values; no lane
    // would do anything else meanwhile).  Walkers on the generic path evaluate the bin with every other one.
    if (run && !any_big && S.t.nfft % 128 == 0) {
        const int ilay0 = sea ? 1 : 0;
        const double omg_nyq = (double)(S.t.nfft / 2) * S.t.domg;
        double2 ur, uz;
        if (sea)
            stage_nyquist<3>(nyq, tl, nl, ilay0, S.t.ipha[f], omg_nyq, ur, uz);
        else
            stage_nyquist<2>(nyq, tl, nl, ilay0, S.t.ipha[f], omg_nyq, ur, uz);
        if (sl == 0 && q <= 0) {
            tl[18] = ur.x; tl[19] = ur.y; tl[20] = uz.x; tl[21] = uz.y;
        }
    }
    __syncthreads();
    if (run && q <= 0) {
        // the walker constants to the global image: one coalesced write per group ([17]: the direct-arrival time)
        double *tail = S.gtail + (size_t)bf * GTAIL;
        const int i0 = S.t.sdep > 0.0 ? 1 : 0;
        if (sl == 0) tl[17] = arrival_sum(nl - 1 - i0, terms);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = sl; i < 22; i += G) tail[i] = tl[i];
        if (sl == 0) S.gflag[bf] = (sea ? 1 : 0) | (any_big ? 2 : 0);
    }
}

template <int G>
static void launch_stage_g(const StageParams &S, unsigned nbf, int nlay_pad, hipStream_t s)
{
    constexpr unsigned NG = 64 / G;
    StageParams P = S;
    // tiny batches of shallow contexts (the per-call drop-in, a handful of chains): a wave per item, its four groups
    // one part each -- half the latency; from ~1000 items on the packed layout is faster again (C2: 2 %)
    P.spread = (G == 16 && nbf <= 256) ? 1 : 0;
    const unsigned nwave = P.spread ? nbf : (nbf + NG - 1) / NG;
    // small batches: one wave per block, so that the few waves spread over the CUs; deep contexts: as many waves as
    // 60 KB of LDS rows allow
    const size_t wave_bytes = sizeof(double) * NG * stage_row_doubles(nlay_pad);
    unsigned wpb = nwave <= 2048 ? 1 : 4;
    while (wpb > 1 && wpb * wave_bytes > 60 * 1024) wpb >>= 1;
    static LdsOptIn opt;     // one wave of a 200-layer context needs 18 KB per group
    opt(reinterpret_cast<const void *>(stage_kernel<G>));
    hipLaunchKernelGGL(stage_kernel<G>, dim3((nwave + wpb - 1) / wpb), dim3(64 * wpb), wpb * wave_bytes, s, P);
}

void launch_stage(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s)
{
    StageParams S{t, b, w.gcoef, w.gtail, w.gflag, 0, w.item_state, w.err, w.nslots};
    const unsigned nbf = (unsigned)(b.nb * t.nfwd);
    // lanes per (item, trace): 16 up to 32 layers (a walker of more than 16 makes two passes over its lanes; with the
    // batch in depth order a wave's four walkers are alike, and the mean walker is half as deep as the deepest the
    // context allows), 32 or 64 beyond
    const int nsolid_max = b.nlay_pad - 1;
    if (nsolid_max <= 32)
        launch_stage_g<16>(S, nbf, b.nlay_pad, s);
    else if (nsolid_max <= 64)
        launch_stage_g<32>(S, nbf, b.nlay_pad, s);
    else
        launch_stage_g<64>(S, nbf, b.nlay_pad, s);
}

template <int BK, int NCOL>
static void launch_spectra_one(dim3 grid, dim3 block, size_t lds, hipStream_t s, const SpectraParams &P)
{
    static LdsOptIn opt;     // nlay_max 200 (the reference's limit) needs 40 KB; deeper contexts more than 64
    opt(reinterpret_cast<const void *>(spectra_kernel<BK, NCOL>));
    hipLaunchKernelGGL((spectra_kernel<BK, NCOL>), grid, block, lds, s, P);
}

template <int NCOL>
static void launch_spectra_ncol(int chain, dim3 grid, dim3 block, size_t lds, hipStream_t s, const SpectraParams &P)
{
    switch (chain) {
    case 2: launch_spectra_one<2, NCOL>(grid, block, lds, s, P); break;
    case 3: launch_spectra_one<3, NCOL>(grid, block, lds, s, P); break;
    case 4: launch_spectra_one<4, NCOL>(grid, block, lds, s, P); break;
    case 8: launch_spectra_one<8, NCOL>(grid, block, lds, s, P); break;
    default: launch_spectra_one<0, NCOL>(grid, block, lds, s, P); break;
    }
}

// chain = bins per phase chain (0 / 1: every phase by a full sincos; 2, 3, 4, 8)
void launch_spectra(const DeviceTables &t, const BatchArgs &b, double2 *spec, int nsplit, int chain,
                    int waves_per_block, int *slow_list, int *slow_count, const WalkerState &w, hipStream_t s)
{
    SpectraParams P{t, b, spec, nsplit, slow_list, slow_count, w};
    int wpb = waves_per_block < 1 ? 1 : (waves_per_block > 4 ? 4 : waves_per_block);
    if (wpb > nsplit) wpb = nsplit;
    const int nblk = (nsplit + wpb - 1) / wpb;
    const dim3 grid((unsigned)(b.nb * t.nfwd * nblk)), block(64 * wpb);
    const size_t lds = spectra_lds_bytes(b.nlay_pad);
    if (t.sdep > 0.0)
        launch_spectra_ncol<3>(chain, grid, block, lds, s, P);
    else
        launch_spectra_ncol<2>(chain, grid, block, lds, s, P);
    if (chain == 2 || chain == 3 || chain == 4 || chain == 8) return;   // generic path handled in place
    static LdsOptIn opt_slow;
    opt_slow(reinterpret_cast<const void *>(spectra_slow_kernel));
    hipLaunchKernelGGL(spectra_slow_kernel, dim3(512), dim3(64), lds, s, P);
}

// ---------------------------------------------------------------------------
// K2  trace: decon / filter / c2r / shift / normalise / misfit quadratic form
// ---------------------------------------------------------------------------
constexpr int TRACE_THREADS = 256;

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide max over TRACE_THREADS threads; red needs >= 4 doubles
__device__ __forceinline__ double block_max(double v, double *red)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Fortran nint: round half away from zero
__device__ __forceinline__ int f_nint(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : -floor(0.5 - x)); }

// the reference's mod(x, n) of the shift maps (forward.f90:179,188) for in-range shifts: into [0, n)
__device__ __forceinline__ int wrap_index(int x, int n)
{
    if ((n & (n - 1)) == 0) return x & (n - 1);
    const int r = x % n;
    return r < 0 ? r + n : r;
}

__device__ __noinline__ int calc_npre(double t_start, double tp, double delta, int ipha)
{
#pragma clang fp contract(off)
    // forward.f90:177 / :186
    const double num = ipha == 1 ? (-t_start - tp) : (-t_start + tp);
    return f_nint(num / delta);
}

// ---------------------------------------------------------------------------
// In-LDS inverse complex FFT (sign +, unnormalised) used for the c2r step.
// Mixed radix, decimation in time, in place: the input is written in digit-reversed
// order, pass p (radix R_p, stride s_p = R_0 ... R_{p-1}) combines R_p sub-transforms
// of length s_p; every pass is one register-resident radix-R butterfly per thread
// (R <= 16), so n = 4096 needs 3 passes / 3 barriers instead of the 12 of radix 2.
// LDS index i is padded to i + (i >> 4) + (i >> 8): the (i >> 4) term keeps the stride-1
// pass (16 contiguous elements per lane) off a single bank, the (i >> 8) term does the
// same for the digit-reversed fill (consecutive bins land 256 elements apart; without it
// rocprof showed 57 % of the kernel's LDS cycles as bank conflicts, all from those writes).
// ---------------------------------------------------------------------------
constexpr int FFT_MAX_PASSES = 4;

struct FftPlan {
    int npass;
    int radix_log2[FFT_MAX_PASSES];  // execution order; stride of pass p = prod of earlier radices
};

__host__ __device__ inline int fft_pad(int i) { return i + (i >> 4) + (i >> 8); }

// position (unpadded) of input bin k in the digit-reversed DIT layout
__device__ __forceinline__ int fft_input_pos(const FftPlan &pl, int log2n, int k)
{
    int pos = 0, span_log2 = log2n;
#pragma unroll
    for (int p = FFT_MAX_PASSES - 1; p >= 0; --p) {
        if (p < pl.npass) {
            const int rl = pl.radix_log2[p];
            const int d = k & ((1 << rl) - 1);
            k >>= rl;
            span_log2 -= rl;
            pos += d << span_log2;
        }
    }
    return pos;
}

// v *= exp(+2 pi i e / 16), e = 0..7 compile-time after unrolling
__device__ __forceinline__ double2 mul_w16(double2 v, int e)
{
    constexpr double C = 0.92387953251128673848;  // cos(pi/8)
    constexpr double S = 0.38268343236508978178;  // sin(pi/8)
    constexpr double H = 0.70710678118654752440;  // sqrt(1/2)
    switch (e) {
    case 0: return v;
    case 1: return make_double2(v.x * C - v.y * S, v.x * S + v.y * C);
    case 2: return make_double2((v.x - v.y) * H, (v.x + v.y) * H);
    case 3: return make_double2(v.x * S - v.y * C, v.x * C + v.y * S);
    case 4: return make_double2(-v.y, v.x);
    case 5: return make_double2(-v.x * S - v.y * C, v.x * C - v.y * S);
    case 6: return make_double2((-v.x - v.y) * H, (v.x - v.y) * H);
    default: return make_double2(-v.x * C - v.y * S, v.x * S - v.y * C);
    }
}

template <int LOG2R>
__device__ __forceinline__ int bitrev_small(int i)
{
    int r = 0;
#pragma unroll
    for (int b = 0; b < LOG2R; ++b) r |= ((i >> b) & 1) << (LOG2R - 1 - b);
    return r;
}

// radix-R inverse DFT of v[0..R) in registers (decimation in frequency); the result
// for output index j sits in v[bitrev(j)].
template <int LOG2R>
__device__ __forceinline__ void dft_regs(double2 (&v)[1 << LOG2R])
{
    constexpr int R = 1 << LOG2R;
#pragma unroll
    for (int sl = LOG2R - 1; sl >= 0; --sl) {
        const int sz = 1 << sl;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if ((i & sz) == 0) {
                const int j = i + sz;
                const double2 a0 = v[i], a1 = v[j];
                v[i] = cadd(a0, a1);
                v[j] = mul_w16(csub(a0, a1), (i & (sz - 1)) * (8 >> sl));
            }
        }
    }
}

// (tw_sh: the table holds exp(+2 pi i k / (n << tw_sh)) -- a longer transform's table read with a stride; 0 in the
// in-LDS transforms of the trace kernels)
template <int LOG2R, int THREADS>
__device__ __forceinline__ void fft_pass(double2 *a, int log2n, int stride_log2, const double2 *__restrict__ tw,
                                         int tid, int tw_sh = 0)
{
    constexpr int R = 1 << LOG2R;
    const int n = 1 << log2n;
    const int stride = 1 << stride_log2;
    const int tw_mul_log2 = log2n - stride_log2 - LOG2R;  // n / (stride * R)
    for (int b = tid; b < (n >> LOG2R); b += THREADS) {
        const int jp = b & (stride - 1);
        const int base = ((b >> stride_log2) << (stride_log2 + LOG2R)) + jp;
        double2 v[R], w[R];
        // issue every twiddle load (L2-resident table) before the LDS reads so that the
        // whole pass pays one global round trip, not R-1 of them
        if (stride_log2 > 0) {
#pragma unroll
            for (int k = 1; k < R; ++k) w[k] = tw[(unsigned)(((jp * k) << tw_mul_log2) << tw_sh)];   // (< n: the table holds the full turn)
        }
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = a[fft_pad(base + (k << stride_log2))];
        if (stride_log2 > 0) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = cmul(v[k], w[k]);
        }
        dft_regs<LOG2R>(v);
#pragma unroll
        for (int k = 0; k < R; ++k) a[fft_pad(base + (bitrev_small<LOG2R>(k) << stride_log2))] = v[k];
    }
}

// n = 4096 = 16^3 with 256 threads: every pass is one radix-16 butterfly per thread.  The twiddles of
// passes 2 and 3 depend only on the thread index, so their 30 loads (L2-resident table) are issued
// before pass 1 and land while it runs; the last pass leaves its 16 outputs in registers:
// v[k] is sample tid + (bitrev4(k) << 8).  Same operations as fft_pass, pass by pass.
// the fifteen twiddles w^1 .. w^15 of a radix-16 butterfly from w^1, w^2, w^4, w^8 (each product within ~4 ulp of the
// table's value): four loads per pass and thread instead of fifteen -- the table is L2-resident, but at the C4 shape
// the fifteen-load version moved 3 GB per launch through the vector memory path for twiddles alone
__device__ __forceinline__ void tw16_expand(double2 (&w)[16])
{
    w[3] = cmul(w[2], w[1]);
    w[5] = cmul(w[4], w[1]);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
#pragma unroll
    for (int k = 1; k < 8; ++k) w[8 + k] = cmul(w[8], w[k]);
}

__device__ __forceinline__ void fft4096_regs(double2 *a, const double2 *__restrict__ tw, int tid, double2 (&v)[16],
                                             int tw_sh = 0)
{
    double2 w2[16], w3[16];
    const int jp = tid & 15;
    // (requested before pass 1, which needs none: they land while it runs)
#pragma unroll
    for (int k = 1; k < 16; k <<= 1) w2[k] = tw[(unsigned)(((jp * k) << 4) << tw_sh)];
#pragma unroll
    for (int k = 1; k < 16; k <<= 1) w3[k] = tw[(unsigned)((tid * k) << tw_sh)];
    fft_pass<4, TRACE_THREADS>(a, 12, 0, tw, tid);          // pass 1: stride 1, no twiddles
    __syncthreads();
    {                                                       // pass 2: stride 16
        const int base = ((tid >> 4) << 8) + jp;
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = a[fft_pad(base + (k << 4))];
        tw16_expand(w2);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w2[k]);
        dft_regs<4>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) a[fft_pad(base + (bitrev_small<4>(k) << 4))] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = a[fft_pad(tid + (k << 8))];   // pass 3: stride 256
    tw16_expand(w3);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w3[k]);
    dft_regs<4>(v);
}

// n = 256 * R with R = 2, 4, 8 (nfft 512, 1024, 2048): the passes before the last through LDS as in
// fft_inverse_lds, the last pass -- one radix-R butterfly per thread, stride 256 -- left in registers:
// v[k] is sample tid + (bitrev_R(k) << 8).
template <int LOG2R>
__device__ __forceinline__ void fft_regs_last(double2 *a, const FftPlan &pl, int log2n, const double2 *__restrict__ tw,
                                              int tid, double2 (&v)[1 << LOG2R])
{
    constexpr int R = 1 << LOG2R;
    int stride_log2 = 0;
    for (int p = 0; p + 1 < pl.npass; ++p) {
        fft_pass<4, TRACE_THREADS>(a, log2n, stride_log2, tw, tid);   // these sizes: radix 16 before the last
        stride_log2 += 4;
        __syncthreads();
    }
    double2 w[R];
#pragma unroll
    for (int k = 1; k < R; ++k) w[k] = tw[(unsigned)(tid * k)];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = a[fft_pad(tid + (k << 8))];
#pragma unroll
    for (int k = 1; k < R; ++k) v[k] = cmul(v[k], w[k]);
    dft_regs<LOG2R>(v);
}

template <int THREADS>
__device__ __forceinline__ void fft_inverse_lds(double2 *a, const FftPlan &pl, int log2n,
                                                const double2 *__restrict__ tw, int tid)
{
    int stride_log2 = 0;
    for (int p = 0; p < pl.npass; ++p) {
        switch (pl.radix_log2[p]) {
        case 4: fft_pass<4, THREADS>(a, log2n, stride_log2, tw, tid); break;
        case 3: fft_pass<3, THREADS>(a, log2n, stride_log2, tw, tid); break;
        case 2: fft_pass<2, THREADS>(a, log2n, stride_log2, tw, tid); break;
        default: fft_pass<1, THREADS>(a, log2n, stride_log2, tw, tid); break;
        }
        stride_log2 += pl.radix_log2[p];
        __syncthreads();
    }
}

static FftPlan make_fft_plan(int log2n)
{
    FftPlan pl{};
    int rem = log2n;
    pl.npass = 0;
    while (rem > 0) {
        const int r = rem >= 4 ? 4 : rem;
        pl.radix_log2[pl.npass++] = r;
        rem -= r;
    }
    return pl;
}

// phi = (misfit . R^-1) . misfit (likelihood.f90:92-93), block-wide (TRACE_THREADS).
// phi1(j) = sum_i misfit(i) r_inv(i,j).  r_inv_t is the transposed image, so for a fixed
// row i consecutive lanes (columns j) read consecutive addresses (coalesced, L2-resident)
// and misfit(i) is an LDS broadcast.  The 4 waves take contiguous quarters of the rows
// (ascending i inside each, like the reference's matmul); the quarter sums are combined
// in wave order through `part` ([4][nsmp] doubles of LDS).  Result valid in thread 0.
__device__ __forceinline__ double quad_form(const DeviceTables &t, int itrc, const double *mis, double *part,
                                            double *red, int tid)
{
    // The partition of the rows into four quarters and the summation order are the same in every block shape and in
    // phi_deferred_kernel (bit-identical values).  Eight-wave blocks (fused8_kernel, fusedc_kernel) spread the COLUMNS
    // of a quarter over two waves -- wave q takes columns lane, lane + 128, ..., wave q + 4 columns lane + 64, ... --
    // so the usual 101-sample window is one pass per wave instead of two.  The step is a chain of L2 round trips
    // (80 KB of R^-1 per trace, nothing else of the block runs meanwhile): 16 rows' loads are in flight per trip.
    const int nsmp = t.nsmp;
    const double *__restrict__ RT = t.r_inv_t + (size_t)itrc * nsmp * nsmp;
    const int wv = tid >> 6, lane = tid & 63;
    const int quarter = wv & 3, cpass = wv >> 2, ncpass = (int)(blockDim.x >> 8);   // 1 (256 threads) or 2 (512)
    const int rows = (nsmp + 3) >> 2;
    const int r0 = quarter * rows, r1 = min(nsmp, r0 + rows);
    __syncthreads();                                          // `part` may alias a buffer still being read
    for (int j = lane + 64 * cpass; j < nsmp; j += 64 * ncpass) {
        double acc = 0.0;
#pragma unroll 16
        for (int i = r0; i < r1; ++i) acc = fma(mis[i], RT[(size_t)i * nsmp + j], acc);
        part[quarter * nsmp + j] = acc;
    }
    __syncthreads();
    double acc = 0.0;
    if (wv < 4) {
        for (int j = tid; j < nsmp; j += TRACE_THREADS) {
            const double phi1 = ((part[j] + part[nsmp + j]) + part[2 * nsmp + j]) + part[3 * nsmp + j];
            acc = fma(phi1, mis[j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wv] = acc;
    }
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// the work region holds, at different times, the padded FFT array, the per-layer
// direct-arrival terms and the 4 x nsmp quarter sums of the quadratic form
__host__ __device__ inline size_t trace_work_doubles(int nfft, int nsmp, int nlay_pad)
{
    size_t d = 2 * (size_t)fft_pad(nfft);
    if ((size_t)4 * nsmp > d) d = (size_t)4 * nsmp;
    if ((size_t)nlay_pad > d) d = (size_t)nlay_pad;
    return (d + 1) & ~(size_t)1;
}

// log-likelihood of one batch item from its per-trace quadratic forms
// (likelihood.f90:86,94-96; same operation order, no FMA contraction).  `uncached`
// reads phi with agent-scope loads: other blocks (other CUs) produced the values.
__device__ __noinline__ double logl_from_phi(const double *phi, const double *sig, int ntrc, int nsmp, bool uncached)
{
#pragma clang fp contract(off)
    double ll = 0.0;
    for (int it = 0; it < ntrc; ++it) {
        const double ph = uncached ? __hip_atomic_load(phi + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : phi[it];
        const double sg = sig[it];
        const double q = 0.5 * ph / (sg * sg);
        const double r = (double)nsmp * log(sg);
        ll = ll - q - r;
    }
    return ll;
}

// What a trace kernel does with a batch item that is NOT to be evaluated.  item_state[ib] is written for every item of
// a batch by stage_kernel, which also checks the item (StageParams): 1 evaluate; 0 sigma-only proposal -- the stored
// trace is re-used (likelihood.f90:81), so is its cached quadratic form; -1 skipped (fwd_flag < 0: a null proposal, an
// invalid model from rf_eval_models): logL = NaN; -2 refused by the input check (nlay / walker id / fwd_flag out of
// range): logL = NaN and no walker state is touched.  `lead`: the one thread of the item that reports.
__device__ __forceinline__ bool item_not_evaluated(const BatchArgs &b, const WalkerState &w, const DeviceTables &t, int ib,
                                                   bool lead)
{
    const int st = w.item_state[ib];
    if (st == 1) return false;
    if (lead) {
        if (st == 0) {
            const int wk = b.walker_ids[ib];
            const double *phi = w.phi + ((size_t)w.cur_slot[wk] * w.nslots + wk) * t.ntrc;
            b.logl[ib] = logl_from_phi(phi, b.sig + (size_t)ib * t.ntrc, t.ntrc, t.nsmp, false);
            w.prop_fwd[wk] = 0;
        } else {
            b.logl[ib] = __longlong_as_double(0x7ff8000000000000LL);
            if (st == -1) w.prop_fwd[b.walker_ids[ib]] = 0;
        }
    }
    return true;
}

struct TraceParams {
    DeviceTables t;
    BatchArgs b;
    const double2 *spec;
    WalkerState w;
    int log2n;
    FftPlan plan;
    int *slow_count;   // re-armed here for the next batch (the slow kernel ran earlier on the stream)
    int ablate;        // RFGPU_DIAGNOSTICS builds only: stop the tail after phase N (timing split, results invalid)
    int defer_logl;    // 1: misfits to HBM; quadratic form + logL by phi_deferred_kernel after this launch
                       // 2: the same for phi_gemm_kernel (long windows: no misfits in LDS at all, t.lds_nsmp = 0)
    double *extra_out; // nullptr, or [ntrc][nfft] (device-mapped host memory): second copy of the proposed trace of
                       // batch item 0 -- the per-call drop-in gets prop_rft without a gather kernel
    double2 *xbuf;     // nullptr, or [nslots * ntrc][fft_pad(nfft)]: the time series of trace_anyn_kernel<true>
};

constexpr int PHI_W = 8;   // batch items per block of phi_deferred_kernel
constexpr int PHI_ROWS = 13;   // rows of R^-1 whose loads are in flight together

// FFT with the last pass in registers, vertical maximum, shift / normalise / store and misfit for
// nfft = 256 * 2^LOG2R (trace_tail).  Returns true when the block is finished (ablation, or the misfits
// went to HBM for the follow-up kernels).
template <int LOG2R>
__device__ __forceinline__ bool tail_in_registers(const TraceParams &P, double2 *a, double *mis, double *mis_g,
                                                  double *__restrict__ dst, double *xout,
                                                  const double *__restrict__ obs, double *red, int ipha, bool decon,
                                                  double tp, int tid)
{
    constexpr int R = 1 << LOG2R;
    const DeviceTables &t = P.t;
    const int n = t.nfft, nsmp = t.nsmp;
    // (the integer shift -- a call with a division -- and the maximum's LDS cell are prepared ahead of the transform:
    // the tail is a chain of latencies, see w8_fft_store)
    const int npre = calc_npre(t.t_start, tp, t.delta, ipha);
    if (tid == 0) red[0] = -HUGE_VAL;
    double2 v[R];
    if constexpr (LOG2R == 4)
        fft4096_regs(a, t.twiddle, tid, v);
    else
        fft_regs_last<LOG2R>(a, P.plan, P.log2n, t.twiddle, tid, v);
    RFGPU_ABLATE_AT(2, true);
    double fac = 1.0;
    if (!decon) {
        double m = -HUGE_VAL;
#pragma unroll
        for (int k = 0; k < R; ++k) m = fmax(m, v[k].y);
        // maxval(rx) forward.f90:201 across the block: one LDS atomic per wave and ONE barrier (the transform's
        // barriers separate the cell's initialisation from these updates)
        m = wave_max(m);
        if ((tid & 63) == 0) (void)__hip_atomic_fetch_max(&red[0], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        fac = red[0];
    }
    // one reciprocal per thread instead of a division per sample (each sample within 1 ulp of the quotient); it
    // carries the sign of the S map (decon: fac = 1, the product is exact)
    const double rfac = (ipha == 1 ? 1.0 : -1.0) / fac;
    // The reference's maps rft(i) = rx(mod(n - npre + i, n)) (forward.f90:179) and rft(i) = -rx(mod(n + npre - i + 1, n))
    // (:188), index 0 standing for n, inverted: sample j (1-based) of rx lands at i = j + npre (P) or n + npre + 1 - j
    // (S), taken mod n with 0 -> n; 0-based that is (j - 1 + npre) mod n and (npre - j) mod n -- n is a power of
    // two, so the reference's mod is a mask, also for negative arguments.
    const unsigned mask = (unsigned)(n - 1);
    const int at0 = ipha == 1 ? tid + npre : npre - tid - 1;        // sample j = tid + 1
    const int step = ipha == 1 ? 256 : -256;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned i0 = (unsigned)(at0 + step * bitrev_small<LOG2R>(k)) & mask;   // 0-based sample of rft
        const double val = v[k].x * rfac;                            // forward.f90:202 (see rfac)
        // written once, read rarely: keep it out of L2.  ("trace_window": only samples 1 .. nsmp are kept, trace_len = nsmp)
        if (i0 < (unsigned)P.w.trace_len) __builtin_nontemporal_store(val, &dst[i0]);
        if (xout) xout[i0] = val;
        if (i0 < (unsigned)nsmp) {
            const double m = val - obs[i0];                          // likelihood.f90:88
            if (P.defer_logl)
                mis_g[i0] = m;
            else
                mis[i0] = m;
        }
    }
    return P.defer_logl != 0;   // quadratic form and logL: phi_deferred_kernel, after this launch
}

// Everything after Z is in LDS: inverse FFT, vertical max, shift / normalise / store,
// misfit, quadratic form, log-likelihood (defer_logl: up to the misfit, which then goes to HBM for
// the follow-up kernels).  Shared by trace_kernel (Z filled from the spectra in HBM) and
// fused_kernel (Z filled straight from the propagator registers).
__device__ __forceinline__ void trace_tail(const TraceParams &P, double2 *a, double *mis, double *red, int ib,
                                           int itrc, int walker, int ipha, bool decon, double tp, int slot, int tid,
                                           bool transformed = false)
{
    const DeviceTables &t = P.t;
    const int n = t.nfft, nsmp = t.nsmp;
    RFGPU_ABLATE_AT(1, );
    double *__restrict__ dst =
        P.w.rft + (((size_t)slot * P.w.nslots + walker) * t.ntrc + itrc) * (size_t)P.w.trace_len;
    const double *__restrict__ obs = t.obs + (size_t)itrc * nsmp;
    double *__restrict__ mis_g = P.w.misfit + ((size_t)ib * t.ntrc + itrc) * t.mis_stride;   // defer mode only
    double *xout = (P.extra_out && ib == 0) ? P.extra_out + (size_t)itrc * n : nullptr;
    if (TRACE_THREADS == 256 && n >= 512 && n <= 4096 && !transformed) {
        // nfft 512 .. 4096: the last pass has exactly one butterfly (radix nfft / 256) per thread; its
        // outputs stay in registers for the vertical maximum, the shift and the store -- one LDS write
        // pass, two LDS read passes and a barrier less than the general path below.  Same values, same
        // operations.
        bool done = false;
        switch (P.log2n) {
        case 12: done = tail_in_registers<4>(P, a, mis, mis_g, dst, xout, obs, red, ipha, decon, tp, tid); break;
        case 11: done = tail_in_registers<3>(P, a, mis, mis_g, dst, xout, obs, red, ipha, decon, tp, tid); break;
        case 10: done = tail_in_registers<2>(P, a, mis, mis_g, dst, xout, obs, red, ipha, decon, tp, tid); break;
        default: done = tail_in_registers<1>(P, a, mis, mis_g, dst, xout, obs, red, ipha, decon, tp, tid); break;
        }
        if (done) return;   // ablation / deferred quadratic form
        __syncthreads();
    } else {
    // ---- in-place mixed-radix inverse FFT, sign +, unnormalised (FFTW c2r definition) ---
    // (transformed: `a` already holds the time series -- the direct DFT of trace_anyn_kernel)
    if (!transformed) fft_inverse_lds<TRACE_THREADS>(a, P.plan, P.log2n, t.twiddle, tid);
    // a[fft_pad(j)].x = rx (RF trace), .y = vertical trace
    RFGPU_ABLATE_AT(2, );

    double fac = 1.0;
    if (!decon) {
        double m = -HUGE_VAL;
        for (int j = tid; j < n; j += TRACE_THREADS) m = fmax(m, a[fft_pad(j)].y);
        fac = block_max(m, red);                                     // maxval(rx) forward.f90:201
    }

    // ---- time shift (+ reverse/negate for S), normalise, store, misfit ----------
    const int npre = calc_npre(t.t_start, tp, t.delta, ipha);
    // one reciprocal per thread instead of a division per sample (each sample within 1 ulp of the quotient)
    const double rfac = 1.0 / fac;
    for (int i = tid + 1; i <= n; i += TRACE_THREADS) {
        int j;
        double val;
        if (ipha == 1) {
            j = wrap_index(n - npre + i, n);                         // forward.f90:179
            if (j == 0) j = n;
            val = a[fft_pad(j - 1)].x;
        } else {
            j = wrap_index(n + npre - i + 1, n);                     // forward.f90:188
            if (j == 0) j = n;
            val = -a[fft_pad(j - 1)].x;
        }
        if (!decon) val = val * rfac;                                // forward.f90:202 (see rfac)
        if (i <= P.w.trace_len) dst[i - 1] = val;
        if (xout) xout[i - 1] = val;
        if (i <= nsmp) {
            const double m = val - obs[i - 1];                       // likelihood.f90:88
            if (P.defer_logl)
                mis_g[i - 1] = m;
            else
                mis[i - 1] = m;
        }
    }
    if (P.defer_logl) return;
    __syncthreads();
    }

    RFGPU_ABLATE_AT(3, );
    // ---- phi = (misfit . R^-1) . misfit   (likelihood.f90:92-93) -----------------
    const double phi = quad_form(t, itrc, mis, reinterpret_cast<double *>(a), red, tid);
    RFGPU_ABLATE_AT(4, );
    if (tid == 0) {
        double *phis = P.w.phi + ((size_t)slot * P.w.nslots + walker) * t.ntrc;
        // ---- log-likelihood (likelihood.f90:94-96) by the block that finishes the walker's
        // last trace.  Hand-off of the 8-byte phi values between blocks with agent-scope
        // atomics on both sides (write-through store, drained, then the counter; the last
        // arriver reads with agent-scope loads) -- no release fence, which would write back
        // the whole L2 slice of freshly written traces (measured: 2.6x on this kernel).
        bool last = true;
        if (t.ntrc > 1) {
            __hip_atomic_store(phis + itrc, phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = atomicAdd(P.w.done + ib, 1) == t.ntrc - 1;
            if (last) P.w.done[ib] = 0;
        } else {
            phis[itrc] = phi;
        }
        if (last) {
            P.b.logl[ib] = logl_from_phi(phis, P.b.sig + (size_t)ib * t.ntrc, t.ntrc, t.nsmp, t.ntrc > 1);
            P.w.prop_fwd[walker] = 1;
        }
    }
}

__global__ __launch_bounds__(TRACE_THREADS) void trace_kernel(TraceParams P)
{
    extern __shared__ double2 lds2[];
    const DeviceTables &t = P.t;
    const int n = t.nfft, nh = t.nh;
    double2 *a = lds2;                                   // [fft_pad(n)] FFT work array (padded index)
    double *mis = reinterpret_cast<double *>(a) + trace_work_doubles(n, t.lds_nsmp, P.b.nlay_pad); // [nsmp] misfits
    double *red = mis + ((t.lds_nsmp + 1) & ~1);         // [8] reductions / broadcasts

    const int tid = threadIdx.x;
    const int itrc = blockIdx.x % t.ntrc;
    const int ib = P.b.order ? P.b.order[blockIdx.x / t.ntrc] : blockIdx.x / t.ntrc;
    if (blockIdx.x == 0 && tid == 0) *P.slow_count = 0;
    if (item_not_evaluated(P.b, P.w, t, ib, itrc == 0 && tid == 0)) return;
    const int walker = P.b.walker_ids[ib];
    const int f = t.ray_common ? 0 : itrc;
    const int ipha = t.ipha[itrc];
    const double2 *__restrict__ sr = P.spec + ((size_t)(ib * t.nfwd + f) * 2) * nh; // freq_r
    const double2 *__restrict__ sv = sr + nh;                                        // freq_v
    const double *__restrict__ flt = t.flt + (size_t)itrc * nh;

    // ---- which spectrum becomes the RF, and the direct-arrival time ------------
    const bool decon = t.deconv_mode == 1;
    double wlvl = 0.0;
    const double2 *num = (ipha == 1) ? sr : sv;      // forward.f90:148-163
    const double2 *den = (ipha == 1) ? sv : sr;      // decon only
    if (decon) {
        double m = -HUGE_VAL;
#pragma unroll 4
        for (int k = tid; k < nh; k += TRACE_THREADS) {
            const double2 x = den[k];
            m = fmax(m, x.x * x.x + x.y * x.y);      // forward.f90:458
        }
        wlvl = 0.001 * block_max(m, red);            // forward.f90:460, pcnt = 0.001 (:149)
    }
    // direct-arrival time: stage_kernel; destination = the half that does not hold the walker's current trace
    const double tp = decon ? 0.0 : P.w.gtail[(size_t)(ib * t.nfwd + f) * GTAIL + 17];
    const int slot = 1 - P.w.cur_slot[walker];

    // ---- Z = RF*flt + i * (V*flt), Hermitian-extended, written digit-reversed ----
    // The spectra come from HBM: each thread first issues the loads of FILL_CHUNK bins
    // (coalesced, 16 B per lane) and only then touches LDS, so a chunk costs one memory
    // round trip instead of one per bin.
    constexpr int FILL_CHUNK = 4;
    for (int k0 = tid; k0 < nh; k0 += TRACE_THREADS * FILL_CHUNK) {
        double2 rr[FILL_CHUNK], xx[FILL_CHUNK];
        double ff[FILL_CHUNK];
#pragma unroll
        for (int c = 0; c < FILL_CHUNK; ++c) {
            const int k = k0 + c * TRACE_THREADS;
            const int kc = k < nh ? k : nh - 1;
            rr[c] = num[kc];
            xx[c] = decon ? den[kc] : sv[kc];   // decon: denominator; else: vertical spectrum
            ff[c] = flt[kc];
        }
#pragma unroll
        for (int c = 0; c < FILL_CHUNK; ++c) {
            const int k = k0 + c * TRACE_THREADS;
            if (k < nh) {
                double2 r = rr[c];
                double2 V = make_double2(0.0, 0.0);
                const double fk = ff[c];
                if (decon) {
                    const double2 x = xx[c];
                    const double amp = x.x * x.x + x.y * x.y;
                    const double dd = fmax(amp, wlvl);                   // forward.f90:464
                    const double2 yx = cmul(r, make_double2(x.x, -x.y));
                    r = make_double2(yx.x / dd, yx.y / dd);
                } else {
                    V = make_double2(xx[c].x * fk, xx[c].y * fk);        // forward.f90:198
                }
                const double2 R = make_double2(r.x * fk, r.y * fk);      // forward.f90:168
                const int pk = fft_pad(fft_input_pos(P.plan, P.log2n, k));
                if (k == 0 || 2 * k == n) {
                    // c2r ignores Im of the DC and Nyquist bins
                    a[pk] = make_double2(R.x, V.x);
                } else {
                    a[pk] = make_double2(R.x - V.y, R.y + V.x);
                    a[fft_pad(fft_input_pos(P.plan, P.log2n, n - k))] = make_double2(R.x + V.y, V.x - R.y);
                }
            }
        }
    }
    __syncthreads();

    trace_tail(P, a, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tid);
}

// ---------------------------------------------------------------------------
// trace_anyn_kernel: K2 for an nfft that is NOT a power of two (FFTW plans any length, src/fftw.f90:44;
// the reference accepts any nfft, src/params.f90:179).  Same steps as trace_kernel; the c2r step is the
// DEFINITION of the unnormalised inverse real DFT (Hermitian extension of bins 0 .. n/2, imaginary parts
// of the DC and -- for even n -- Nyquist bins ignored) summed directly from a twiddle table:
// O(n^2) per trace instead of O(n log n), which is fine for the rare sizes that need it (n = 1000:
// ~2e6 FMA per trace) and keeps every other stage shared.  Split launch plan only.
// BIG = false: time series, filtered spectra and twiddle table in LDS (nfft up to ~3300).
// BIG = true (longer series, up to ~9000): only the filtered spectra stay in LDS; the twiddles are read from the
// (L2-resident) global table and the time series goes through a global scratch row per block -- same sums in the
// same order, so both variants and every other stage produce the same values.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t anyn_spec_offset(int nfft, int nsmp, int nlay_pad)
{
    return trace_work_doubles(nfft, nsmp, nlay_pad) + (size_t)(((nsmp + 1) & ~1) + 8);   // doubles: after a | mis | red
}

size_t trace_anyn_lds_bytes(int nfft, int nsmp, int nlay_pad)
{
    const size_t nh = (size_t)nfft / 2 + 1;
    return sizeof(double) * anyn_spec_offset(nfft, nsmp, nlay_pad) + sizeof(double2) * (2 * nh + (size_t)nfft);
}

size_t trace_anyn_big_lds_bytes(int nfft, int nsmp)
{
    const size_t nh = (size_t)nfft / 2 + 1;
    return sizeof(double) * (size_t)(((nsmp + 1) & ~1) + 8) + sizeof(double2) * 2 * nh;   // mis | red | zr | zv
}

__host__ __device__ size_t trace_anyn_scratch_entries(int nfft, int nsmp)   // double2 per block: the padded time series, or the quadratic form's [4][nsmp]
{
    const size_t e = (size_t)fft_pad(nfft) + 1;
    return e > (size_t)2 * nsmp ? e : (size_t)2 * nsmp;
}

template <bool BIG>
__global__ __launch_bounds__(TRACE_THREADS) void trace_anyn_kernel(TraceParams P)
{
    extern __shared__ double2 lds2[];
    const DeviceTables &t = P.t;
    const int n = t.nfft, nh = t.nh, nsmp = t.nsmp;
    double2 *a, *zr;
    double *mis;
    if (BIG) {
        a = P.xbuf + (size_t)blockIdx.x * trace_anyn_scratch_entries(n, nsmp);
        mis = reinterpret_cast<double *>(lds2);
        zr = reinterpret_cast<double2 *>(mis + ((t.lds_nsmp + 1) & ~1) + 8);
    } else {
        a = lds2;
        mis = reinterpret_cast<double *>(a) + trace_work_doubles(n, t.lds_nsmp, P.b.nlay_pad);
        zr = reinterpret_cast<double2 *>(reinterpret_cast<double *>(a) + anyn_spec_offset(n, t.lds_nsmp, P.b.nlay_pad));
    }
    double *red = mis + ((t.lds_nsmp + 1) & ~1);
    double2 *zv = zr + nh;                               // filtered RF and vertical spectra, bins 0 .. nh-1
    double2 *tw_lds = zv + nh;                           // (!BIG) exp(+2 pi i k / n), k = 0 .. n-1
    const double2 *tw = BIG ? t.twiddle_any : tw_lds;

    const int tid = threadIdx.x;
    const int itrc = blockIdx.x % t.ntrc;
    const int ib = P.b.order ? P.b.order[blockIdx.x / t.ntrc] : blockIdx.x / t.ntrc;
    if (blockIdx.x == 0 && tid == 0) *P.slow_count = 0;
    if (item_not_evaluated(P.b, P.w, t, ib, itrc == 0 && tid == 0)) return;
    const int walker = P.b.walker_ids[ib];
    const int f = t.ray_common ? 0 : itrc;
    const int ipha = t.ipha[itrc];
    const double2 *__restrict__ sr = P.spec + ((size_t)(ib * t.nfwd + f) * 2) * nh;   // freq_r
    const double2 *__restrict__ sv = sr + nh;                                          // freq_v
    const double *__restrict__ flt = t.flt + (size_t)itrc * nh;
    const bool decon = t.deconv_mode == 1;
    const double2 *num = (ipha == 1) ? sr : sv;      // forward.f90:148-163
    const double2 *den = (ipha == 1) ? sv : sr;      // decon only
    double wlvl = 0.0;
    if (decon) {
        double m = -HUGE_VAL;
        for (int k = tid; k < nh; k += TRACE_THREADS) {
            const double2 x = den[k];
            m = fmax(m, x.x * x.x + x.y * x.y);      // forward.f90:458
        }
        wlvl = 0.001 * block_max(m, red);            // forward.f90:460, pcnt = 0.001 (:149)
    }
    const double tp = decon ? 0.0 : P.w.gtail[(size_t)(ib * t.nfwd + f) * GTAIL + 17];
    const int slot = 1 - P.w.cur_slot[walker];
    if (!BIG)
        for (int k = tid; k < n; k += TRACE_THREADS) tw_lds[k] = t.twiddle_any[k];
    for (int k = tid; k < nh; k += TRACE_THREADS) {
        double2 r = num[k];
        const double fk = flt[k];
        double2 V = make_double2(0.0, 0.0);
        if (decon) {
            const double2 x = den[k];
            const double amp = x.x * x.x + x.y * x.y;
            const double dd = fmax(amp, wlvl);                       // forward.f90:464
            const double2 yx = cmul(r, make_double2(x.x, -x.y));
            r = make_double2(yx.x / dd, yx.y / dd);
        } else {
            V = make_double2(sv[k].x * fk, sv[k].y * fk);            // forward.f90:198
        }
        zr[k] = make_double2(r.x * fk, r.y * fk);                    // forward.f90:168
        zv[k] = V;
    }
    __syncthreads();
    // x[j] = X0 + [n even] (-1)^j X_{n/2} + 2 sum_{0 < k < n/2} Re(X_k e^{+2 pi i j k / n})
    const int kmax = (n - 1) / 2;
    for (int j = tid; j < n; j += TRACE_THREADS) {
        double xr = zr[0].x, xv = zv[0].x;
        if ((n & 1) == 0) {
            const double sg = (j & 1) ? -1.0 : 1.0;
            xr = fma(sg, zr[n / 2].x, xr);
            xv = fma(sg, zv[n / 2].x, xv);
        }
        double sr2 = 0.0, sv2 = 0.0;
        int idx = 0;
        for (int k = 1; k <= kmax; ++k) {
            idx += j;
            if (idx >= n) idx -= n;
            const double2 w = tw[idx], R = zr[k], V = zv[k];
            sr2 += R.x * w.x - R.y * w.y;
            sv2 += V.x * w.x - V.y * w.y;
        }
        a[fft_pad(j)] = make_double2(fma(2.0, sr2, xr), fma(2.0, sv2, xv));
    }
    __syncthreads();
    trace_tail(P, a, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tid, true);
}

// ---------------------------------------------------------------------------
// trace_long_kernel: K2 for the series the in-LDS transforms do not reach -- a power of two beyond 8192 (up to
// 65536), and any other length beyond the direct DFT's sensible range (2048 < nfft <= 32768; FFTW plans any length,
// src/fftw.f90:44).  Same steps as trace_kernel; only the c2r step differs.  Split launch plan.
//
// (a) M = 4096 n2 a power of two, n2 = 2 .. 16: the four-step transform.  With k = n2 k1 + r and j = j1 + 4096 j2,
//       x[j1 + 4096 j2] = sum_r w_n2^(j2 r) [ w_M^(j1 r) Y_r[j1] ],     Y_r = IDFT_4096( Z[n2 k1 + r] over k1 ):
//     stage 1: n2 in-LDS 4096-point transforms (the trace kernels' radix-16 passes, last pass in registers), each
//     output times its twiddle w_M^(j1 r), to a global scratch row Y[r][j1] (coalesced both ways);
//     stage 2: one radix-n2 butterfly per j1 in registers.
// (b) any other nfft = n: Bluestein.  jk = (j^2 + k^2 - (j - k)^2) / 2, so with the chirp c[m] = exp(+i pi m^2 / n)
//       x[j] = c[j] sum_k (Z[k] c[k]) conj(c[j - k]):
//     a linear convolution, done as a circular one of length M >= 2n - 1 (M a power of two: (a) twice):
//     U = DFT_M(Z c, zero-padded), V = U Bhat (Bhat = DFT_M of the wrapped conj chirp: a host table, long double),
//     x[j] = c[j] IDFT_M(V)[j] / M.  The forward transform is conj(IDFT(conj .)).  The chirp's argument is reduced
//     exactly (m^2 mod 2n in integers) on the host.
// The time series (RF trace in .x, vertical trace in .y: both real transforms ride one complex one, as everywhere)
// ends up in the block's scratch row and trace_tail does the rest (vertical maximum, shift / normalise / store,
// misfit, quadratic form, logL).  Blocks are persistent: a grid of at most 2 per CU walks the (item, trace) units,
// so the scratch is two rows of M (+ padding) per resident block, not per trace.  (LongTables: rfgpu_internal.h)
// ---------------------------------------------------------------------------
size_t long_row_entries(int nfft, int m, int nsmp)
{
    size_t e = (size_t)m;
    if ((size_t)fft_pad(nfft) + 1 > e) e = (size_t)fft_pad(nfft) + 1;
    if ((size_t)2 * nsmp > e) e = (size_t)2 * nsmp;
    return (e + 1) & ~(size_t)1;
}

// radix-2^LOG2R inverse butterfly over the n2 sub-transforms of one j1: y[r] -> x[j1 + 4096 j2] for j2 = 0 .. n2 - 1,
// handed to out(j, value)
template <int LOG2R, class Out>
__device__ __forceinline__ void long_stage2(const double2 *__restrict__ Y, int tid, const Out &out)
{
    constexpr int R = 1 << LOG2R;
    for (int j1 = tid; j1 < 4096; j1 += TRACE_THREADS) {
        double2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = Y[(size_t)r * 4096 + j1];
        dft_regs<LOG2R>(v);                       // output j2 sits in v[bitrev(j2)]
#pragma unroll
        for (int k = 0; k < R; ++k) out(j1 + 4096 * bitrev_small<LOG2R>(k), v[k]);
    }
}

// unnormalised inverse DFT of length M = 4096 << log2n2 of the sequence in(k), k = 0 .. M-1; out(j, x[j]).
// a: the block's padded 4096-point LDS array; Y: a global scratch row of M entries.
template <class In, class Out>
__device__ __forceinline__ void long_idft(const LongTables &L, double2 *a, double2 *__restrict__ Y, const FftPlan &plan,
                                          int tid, const In &in, const Out &out)
{
    const int n2 = 1 << L.log2n2;
    for (int r = 0; r < n2; ++r) {
        __syncthreads();                          // the previous sub-transform's last pass has read the array
#pragma unroll 4
        for (int k1 = tid; k1 < 4096; k1 += TRACE_THREADS) a[fft_pad(fft_input_pos(plan, 12, k1))] = in((k1 << L.log2n2) + r);
        __syncthreads();
        double2 v[16];
        fft4096_regs(a, L.tw_m, tid, v, L.log2n2);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j1 = tid + (bitrev_small<4>(k) << 8);
            const double2 w = L.tw_m[(unsigned)(j1 * r) & (unsigned)(L.m - 1)];     // w_M^(j1 r)
            Y[(size_t)r * 4096 + j1] = cmul(v[k], w);
        }
    }
    __threadfence_block();
    __syncthreads();                              // every Y[r][j1] is written (and visible) before stage 2 gathers them
    switch (L.log2n2) {
    case 1: long_stage2<1>(Y, tid, out); break;
    case 2: long_stage2<2>(Y, tid, out); break;
    case 3: long_stage2<3>(Y, tid, out); break;
    default: long_stage2<4>(Y, tid, out); break;
    }
}

struct LongParams {
    TraceParams tp;
    LongTables L;
};

__global__ __launch_bounds__(TRACE_THREADS) void trace_long_kernel(LongParams Q)
{
    extern __shared__ double2 lds2[];
    const TraceParams &P = Q.tp;
    const LongTables &L = Q.L;
    const DeviceTables &t = P.t;
    const int n = t.nfft, nh = t.nh;
    double2 *a = lds2;                                                   // [fft_pad(4096)] the 4096-point work array
    double *mis = reinterpret_cast<double *>(a + ((fft_pad(4095) + 2) & ~1));   // [nsmp] misfits
    double *red = mis + ((t.lds_nsmp + 1) & ~1);                         // [8] reductions
    const int tid = threadIdx.x;
    double2 *Y = L.scratch + (size_t)blockIdx.x * 2 * L.row_entries;      // stage-1 outputs
    double2 *W = Y + L.row_entries;                                       // Bluestein: V = U Bhat; finally the time series
    if (blockIdx.x == 0 && tid == 0) *P.slow_count = 0;
    const bool decon = t.deconv_mode == 1;
    const int units = P.b.nb * t.ntrc;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        __syncthreads();                                                  // the previous unit is finished with LDS and its rows
        const int itrc = u % t.ntrc;
        const int ib = P.b.order ? P.b.order[u / t.ntrc] : u / t.ntrc;
        if (item_not_evaluated(P.b, P.w, t, ib, itrc == 0 && tid == 0)) continue;
        const int walker = P.b.walker_ids[ib];
        const int f = t.ray_common ? 0 : itrc;
        const int ipha = t.ipha[itrc];
        const double2 *__restrict__ sr = P.spec + ((size_t)(ib * t.nfwd + f) * 2) * nh;   // freq_r
        const double2 *__restrict__ sv = sr + nh;                                          // freq_v
        const double *__restrict__ flt = t.flt + (size_t)itrc * nh;
        const double2 *num = (ipha == 1) ? sr : sv;      // forward.f90:148-163
        const double2 *den = (ipha == 1) ? sv : sr;      // decon only
        double wlvl = 0.0;
        if (decon) {
            double m = -HUGE_VAL;
            for (int k = tid; k < nh; k += TRACE_THREADS) {
                const double2 x = den[k];
                m = fmax(m, x.x * x.x + x.y * x.y);      // forward.f90:458
            }
            wlvl = 0.001 * block_max(m, red);            // forward.f90:460, pcnt = 0.001 (:149)
        }
        const double tp = decon ? 0.0 : P.w.gtail[(size_t)(ib * t.nfwd + f) * GTAIL + 17];
        const int slot = 1 - P.w.cur_slot[walker];
        // Z[k], k = 0 .. n-1: RF flt + i V flt with its Hermitian extension (trace_kernel's fill, bin by bin)
        auto zfull = [&](int k) -> double2 {
            const int kk = 2 * k <= n ? k : n - k;
            double2 r = num[kk];
            double2 V = make_double2(0.0, 0.0);
            const double fk = flt[kk];
            if (decon) {
                const double2 x = den[kk];
                const double amp = x.x * x.x + x.y * x.y;
                const double dd = fmax(amp, wlvl);                       // forward.f90:464
                const double2 yx = cmul(r, make_double2(x.x, -x.y));
                r = make_double2(yx.x / dd, yx.y / dd);
            } else {
                const double2 v = sv[kk];
                V = make_double2(v.x * fk, v.y * fk);                    // forward.f90:198
            }
            const double2 R = make_double2(r.x * fk, r.y * fk);          // forward.f90:168
            if (k == 0 || 2 * k == n) return make_double2(R.x, V.x);     // c2r ignores Im of the DC and Nyquist bins
            return 2 * k < n ? make_double2(R.x - V.y, R.y + V.x) : make_double2(R.x + V.y, V.x - R.y);
        };
        if (!L.bluestein) {
            long_idft(L, a, Y, P.plan, tid, zfull, [&](int j, double2 x) { W[fft_pad(j)] = x; });
        } else {
            // U = DFT_M(Z c) = conj(IDFT_M(conj(Z c)));  V = U Bhat
            long_idft(L, a, Y, P.plan, tid,
                      [&](int k) -> double2 {
                          if (k >= n) return make_double2(0.0, 0.0);
                          const double2 zc = cmul(zfull(k), L.chirp[k]);
                          return make_double2(zc.x, -zc.y);
                      },
                      [&](int j, double2 uc) { W[j] = cmul(make_double2(uc.x, -uc.y), L.bhat[j]); });
            __threadfence_block();
            __syncthreads();
            // x[j] = c[j] IDFT_M(V)[j] / M for j < n.  The time series overwrites W, which the stage-1 fills of this
            // transform have finished reading before its stage 2 writes anything (the barrier inside long_idft)
            const double inv_m = 1.0 / (double)L.m;
            long_idft(L, a, Y, P.plan, tid, [&](int k) -> double2 { return W[k]; },
                      [&](int j, double2 y) {
                          if (j < n) {
                              const double2 x = cmul(y, L.chirp[j]);
                              W[fft_pad(j)] = make_double2(x.x * inv_m, x.y * inv_m);
                          }
                      });
        }
        __threadfence_block();
        __syncthreads();
        trace_tail(P, W, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tid, true);
    }
}

size_t trace_long_lds_bytes(int nsmp)
{
    return sizeof(double2) * (size_t)((fft_pad(4095) + 2) & ~1) + sizeof(double) * (size_t)(((nsmp + 1) & ~1) + 8);
}

void launch_trace_long(const DeviceTables &t, const BatchArgs &b, const double2 *spec, const WalkerState &w, int *slow_count,
                       const LongTables &L, int rows, int defer_logl, hipStream_t s)
{
    LongParams Q{};
    Q.tp = TraceParams{t, b, spec, w, 12, make_fft_plan(12), slow_count, 0, defer_logl, nullptr, nullptr};
    Q.L = L;
    const int units = b.nb * t.ntrc;
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(trace_long_kernel));
    hipLaunchKernelGGL(trace_long_kernel, dim3((unsigned)(units < rows ? units : rows)), dim3(TRACE_THREADS),
                       trace_long_lds_bytes(t.lds_nsmp), s, Q);
}

// ---------------------------------------------------------------------------
// fused_kernel: K1 + K2 in one launch for contexts where every trace has its own
// forward computation (rays not common, or a single trace).  One 256-thread block per
// (walker, trace): its 4 waves propagate a quarter of the bins each and deposit
// Z = RF*flt + i V*flt (or, for water-level deconvolution, the raw numerator /
// denominator pair) straight into the digit-reversed FFT array in LDS -- the spectra never
// travel through HBM -- then the block runs the trace tail.  While one block of a CU is in
// its latency-bound tail the other one is in its FP64-bound propagator phase.
// ---------------------------------------------------------------------------
struct LdsSink {
    double2 *a;                 // padded, digit-reversed FFT array
    double2 *side;              // [2] denominators of the DC and Nyquist bins (decon only)
    const double *__restrict__ flt;
    FftPlan plan;
    int log2n, n, nh, ipha;
    bool decon;
    // n = 4096 with three radix-16 passes (every BASELINE shape): a lane's bins are k = lane + 64 it,
    // so the base-16 digit reversal splits into a per-lane constant and a term of the iteration:
    // pos(k) = (lane & 15) 256 + (lane >> 4) 16  +  64 (it & 3) + (it >> 2)
    bool r16x3;
    int lane_pos, lane_pos_m;   // of the lane and of its mirror lane (64 - lane) & 63
    __device__ __forceinline__ int pos_of(int k) const
    {
        if (!r16x3) return fft_pad(fft_input_pos(plan, log2n, k));
        const int it = k >> 6;
        return fft_pad(lane_pos + ((it & 3) << 6) + (it >> 2));
    }
    __device__ __forceinline__ int pos_of_mirror(int k) const   // position of bin n - k, 0 < k < n / 2
    {
        if (!r16x3) return fft_pad(fft_input_pos(plan, log2n, n - k));
        const int it = (k & 63) ? 63 - (k >> 6) : 64 - (k >> 6);
        return fft_pad(lane_pos_m + ((it & 3) << 6) + (it >> 2));
    }
    // Gaussian filter weight of bin k (forward.f90:168,198), fetched by the caller before the bin's boundary
    // condition is evaluated; deconvolution applies the filter after the water level (no weight here)
    __device__ __forceinline__ double weight(int k) const { return (!decon && k < nh) ? flt[k] : 0.0; }
    __device__ __forceinline__ void operator()(int k, double2 ur, double2 uz, double fk, int = 0) const
    {
        if (k >= nh) return;
        const double2 fr = make_double2(ur.x, -ur.y);    // freq_r = conjg(ur)   forward.f90:145
        const double2 fv = make_double2(-uz.x, uz.y);    // freq_v = -conjg(uz)  forward.f90:146
        const double2 num = ipha == 1 ? fr : fv;         // forward.f90:148-163
        const int pk = pos_of(k);
        const bool self = (k == 0 || 2 * k == n);
        if (decon) {
            // keep numerator and denominator; the water level needs the max over all bins first
            const double2 den = ipha == 1 ? fv : fr;
            a[pk] = num;
            if (self)
                side[k == 0 ? 0 : 1] = den;
            else
                a[pos_of_mirror(k)] = den;
        } else {
            const double2 R = make_double2(num.x * fk, num.y * fk);   // forward.f90:168
            const double2 V = make_double2(fv.x * fk, fv.y * fk);     // forward.f90:198
            if (self) {
                a[pk] = make_double2(R.x, V.x);          // c2r ignores Im of the DC and Nyquist bins
            } else {
                a[pk] = make_double2(R.x - V.y, R.y + V.x);
                a[pos_of_mirror(k)] = make_double2(R.x + V.y, V.x - R.y);
            }
        }
    }
};

size_t trace_lds_bytes(int nfft, int nsmp, int nlay_pad);
static FftPlan make_fft_plan(int log2n);

struct FusedParams {
    SpectraParams sp;   // t, b, nsplit (= waves per block), meta pointers unused
    TraceParams tp;     // t, b, w, log2n, plan, slow_count
    int *order_next;    // nullptr, or: block 0 sorts this batch's items by depth for the NEXT launch
};

__device__ __forceinline__ void order_block(int nb, const int *nlay, const int *fwd_flag, int *order, int *hist,
                                            int *start, int *wave_tot);

size_t fused_lds_bytes(int nfft, int nsmp, int nlay_pad)
{
    return trace_lds_bytes(nfft, nsmp, nlay_pad) + sizeof(double2) * 2 + spectra_lds_bytes(nlay_pad);
}

template <int BK, int NCOL>
__global__ __launch_bounds__(TRACE_THREADS, 2) void fused_kernel(FusedParams F)
{
    extern __shared__ double2 lds2[];
    const TraceParams &P = F.tp;
    const DeviceTables &t = P.t;
    const int n = t.nfft, nh = t.nh;
    double2 *a = lds2;
    double *mis = reinterpret_cast<double *>(a) + trace_work_doubles(n, t.lds_nsmp, P.b.nlay_pad);
    double *red = mis + ((t.lds_nsmp + 1) & ~1);
    double2 *side = reinterpret_cast<double2 *>(red + 8);
    double *coef = reinterpret_cast<double *>(side + 2);
    double *tail = coef + (size_t)P.b.nlay_pad * NCOEF;

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    RFGPU_STAGGER(bid);
    if (F.order_next && bid == P.b.nb * t.ntrc) {
        // Longest-first dispatch order for the next launch of this batch shape, computed by one extra
        // block instead of a separate kernel in front of every launch.  It is the LAST block: it takes the
        // first slot that becomes free near the end of the launch, where the shallowest walkers leave slack
        // (as the first block it delayed one slot's whole sequence and with it the kernel).  The next
        // launch's depths differ from these by a proposal step (+-1 layer for some walkers): the order is
        // then slightly stale, which costs balance, never correctness.
        int *w = reinterpret_cast<int *>(lds2);
        order_block(P.b.nb, P.b.nlay, P.b.fwd_flag, F.order_next, w, w + 256, w + 512);
        return;
    }
    const int itrc = bid % t.ntrc;            // == forward-trace index here (nfwd == ntrc)
    const int ib = P.b.order ? P.b.order[bid / t.ntrc] : bid / t.ntrc;
    if (bid == 0 && tid == 0) *P.slow_count = 0;
    if (item_not_evaluated(P.b, P.w, t, ib, itrc == 0 && tid == 0)) return;
    const int walker = P.b.walker_ids[ib];
    const int ipha = t.ipha[itrc];
    const bool decon = t.deconv_mode == 1;

    // ---- the layer stack's constants and the direct-arrival time: stage_kernel's output.  The fast paths read
    // them through the scalar data path straight from the global image; only a walker on the generic path (rare)
    // copies the image into LDS first
    const int bfi = ib * t.ntrc + itrc;
    const int nl = P.b.nlay[ib];
    const int stage_flags = P.w.gflag[bfi];
    const bool sea = stage_flags & 1;             // beta(1) < 0  (forward.f90:229)
    const bool big = (stage_flags & 2) != 0;
    const int ilay0 = sea ? 1 : 0;
    const bool generic = big || sea != (NCOL == 3);
    const KPtr gcoef = as_scalar_ptr(P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF);
    const KPtr gtail = as_scalar_ptr(P.w.gtail + (size_t)bfi * GTAIL);
    if (generic) {
        int nl2, il2;
        bool sea2;
        (void)load_staged(P.w, P.b, t.ntrc, ib, itrc, coef, tail, nl2, il2, sea2);
    }
    const int slot = 1 - P.w.cur_slot[walker];
    RFGPU_ABLATE_AT(5, );   // timing diagnostics: launch + staging only

    // ---- propagator phase: 4 waves x interleaved chunks of bins -> Z in LDS ----------------
    // Optional (off by default, rf_set_option "bin_cutoff"): bins whose Gaussian filter weight is below
    // cutoff * flt(0) are not propagated; their Z entries are zero.  Only without deconvolution
    // (the water level needs the maximum over every bin).
    const int nh_eff = (t.nh_active && !decon) ? t.nh_active[itrc] : nh;
    SpectraParams sp = F.sp;
    sp.t.nh = nh_eff;
    RFGPU_ABLATE_COPY(sp);
    for (int k = nh_eff + tid; k < nh; k += TRACE_THREADS) {
        a[fft_pad(fft_input_pos(P.plan, P.log2n, k))] = make_double2(0.0, 0.0);
        if (k != 0 && 2 * k != n) a[fft_pad(fft_input_pos(P.plan, P.log2n, n - k))] = make_double2(0.0, 0.0);
    }
    const int wave = tid >> 6, lane = tid & 63;
    const bool r16x3 = P.log2n == 12 && P.plan.npass == 3 && P.plan.radix_log2[0] == 4 && P.plan.radix_log2[1] == 4 &&
                       P.plan.radix_log2[2] == 4;
    const int lane_m = (64 - lane) & 63;
    const LdsSink sink{a, side, t.flt + (size_t)itrc * nh, P.plan, P.log2n, n, nh_eff, ipha, decon, r16x3,
                       ((lane & 15) << 8) + ((lane >> 4) << 4), ((lane_m & 15) << 8) + ((lane_m >> 4) << 4)};
    if (generic) {
        if (sea)
            spectra_body<0, 3, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, wave, lane);
        else
            spectra_body<0, 2, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, wave, lane);
    } else {
        spectra_body<BK, NCOL, true>(sp, gcoef, gtail, nl, ilay0, ipha, sink, wave, lane);
    }
    __syncthreads();
    const double tp = decon ? 0.0 : gtail[17];

    if (decon) {
        // water_level_decon (forward.f90:447-470) in place: slot(k) holds the numerator,
        // slot(n-k) the denominator of bin k
        double m = -HUGE_VAL;
        for (int k = tid; k < nh; k += TRACE_THREADS) {
            const bool self = (k == 0 || 2 * k == n);
            const double2 x = self ? side[k == 0 ? 0 : 1] : a[fft_pad(fft_input_pos(P.plan, P.log2n, n - k))];
            m = fmax(m, x.x * x.x + x.y * x.y);                   // forward.f90:458
        }
        const double wlvl = 0.001 * block_max(m, red);            // forward.f90:460, pcnt = 0.001 (:149)
        const double *__restrict__ flt = t.flt + (size_t)itrc * nh;
        for (int k = tid; k < nh; k += TRACE_THREADS) {
            const bool self = (k == 0 || 2 * k == n);
            const int pk = fft_pad(fft_input_pos(P.plan, P.log2n, k));
            const int pnk = self ? pk : fft_pad(fft_input_pos(P.plan, P.log2n, n - k));
            const double2 y = a[pk];
            const double2 x = self ? side[k == 0 ? 0 : 1] : a[pnk];
            const double amp = x.x * x.x + x.y * x.y;
            const double dd = fmax(amp, wlvl);                    // forward.f90:464
            const double2 yx = cmul(y, make_double2(x.x, -x.y));
            const double fk = flt[k];
            const double2 R = make_double2(yx.x / dd * fk, yx.y / dd * fk);
            if (self) {
                a[pk] = make_double2(R.x, 0.0);
            } else {
                a[pk] = R;
                a[pnk] = make_double2(R.x, -R.y);
            }
        }
        __syncthreads();
    }
    // (the 8-bin ocean chain leaves no register for what the compiler would compute ahead of the propagator phase from
    // the thread index -- shift map, masks, addresses of the tail -- and then spill: its tail gets an opaque copy)
    int tid_tail = tid;
    if constexpr (NCOL != 2 && BK >= 8) asm volatile("" : "+v"(tid_tail));
    trace_tail(P, a, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tid_tail);
}

// ---------------------------------------------------------------------------
// fused8_kernel: the fused kernel for nfft = 4096 on land (every land BASELINE shape) with 512-thread blocks:
// EIGHT waves per (walker, trace), each propagating one 4-bin phase chain (+ the Nyquist iteration on the last),
// then a four-pass radix-8 FFT with one butterfly per thread and pass, the last pass left in registers.
//
// Why: the 256-thread kernel holds two blocks = two waves per SIMD (76 KB of LDS and ~240 VGPRs each); while a
// block is in its latency-bound tail the other computes with ONE wave per SIMD, and a single wave issues an
// fp64 instruction only every ~10 cycles (tools/fp64_peak.hip).  Here the constants of the chained-phase loop
// live in SGPRs (scalar loads from stage_kernel's image), the 4-bin chain needs ~120 VGPRs and the radix-8
// butterflies 32 + 28, so the kernel fits 128 VGPRs: two 8-wave blocks per CU = FOUR waves per SIMD, and a
// block in its tail still leaves two.  A block's tail is also twice as parallel.  LDS: the FFT array only
// (70.7 KB) -- walkers on the generic path copy their constants over the misfit / reduction area's neighbour
// (rare; their code may spill, it is off the hot path).
//
// Same arithmetic as fused_kernel per bin and layer; the FFT factorisation differs (8^4 instead of 16^3), so
// traces differ from the 256-thread kernel's in the last bits (both within 1e-12 of the oracle).
// LDS index padding i + (i >> 4) + (i >> 9): the first term keeps the stride-1 pass (8 contiguous elements per
// lane) conflict-free, the second spreads the digit-reversed fill (consecutive bins land 512 elements apart).
// ---------------------------------------------------------------------------
constexpr int W8_THREADS = 512;
__host__ __device__ inline int w8_pad(int i) { return i + (i >> 4) + (i >> 9); }

// position (unpadded) of bin k in the digit-reversed input order of four radix-8 DIT passes
__device__ __forceinline__ int w8_pos(int k)
{
    return ((k & 7) << 9) | (((k >> 3) & 7) << 6) | (((k >> 6) & 7) << 3) | (k >> 9);
}

struct W8Sink {
    double2 *a;                 // padded, digit-reversed FFT array
    double2 *side;              // [2] denominators of the DC and Nyquist bins (decon only)
    const double *__restrict__ flt;
    int nh, ipha;
    bool decon;
    int lane_pos, lane_pos_m;   // digit-reversed contribution of the lane and of its mirror lane (64 - lane) & 63
    __device__ __forceinline__ int pos_of(int k) const
    {
        const int it = k >> 6;                              // k = lane + 64 it
        return w8_pad(lane_pos + ((it & 7) << 3) + (it >> 3));
    }
    __device__ __forceinline__ int pos_of_mirror(int k) const   // position of bin 4096 - k, 0 < k < 2048
    {
        const int it = (k & 63) ? 63 - (k >> 6) : 64 - (k >> 6);
        return w8_pad(lane_pos_m + ((it & 7) << 3) + (it >> 3));
    }
    // Gaussian filter weight of bin k (forward.f90:168,198), fetched by the caller before the bin's boundary
    // condition is evaluated; deconvolution applies the filter after the water level (no weight here)
    __device__ __forceinline__ double weight(int k) const { return (!decon && k < nh) ? flt[k] : 0.0; }
    __device__ __forceinline__ void operator()(int k, double2 ur, double2 uz, double fk, int = 0) const
    {
        if (k >= nh) return;
        const double2 fr = make_double2(ur.x, -ur.y);    // freq_r = conjg(ur)   forward.f90:145
        const double2 fv = make_double2(-uz.x, uz.y);    // freq_v = -conjg(uz)  forward.f90:146
        const double2 num = ipha == 1 ? fr : fv;         // forward.f90:148-163
        const int pk = pos_of(k);
        const bool self = (k == 0 || k == 2048);
        if (decon) {
            const double2 den = ipha == 1 ? fv : fr;
            a[pk] = num;
            if (self)
                side[k == 0 ? 0 : 1] = den;
            else
                a[pos_of_mirror(k)] = den;
        } else {
            const double2 R = make_double2(num.x * fk, num.y * fk);   // forward.f90:168
            const double2 V = make_double2(fv.x * fk, fv.y * fk);     // forward.f90:198
            if (self) {
                a[pk] = make_double2(R.x, V.x);          // c2r ignores Im of the DC and Nyquist bins
            } else {
                a[pk] = make_double2(R.x - V.y, R.y + V.x);
                a[pos_of_mirror(k)] = make_double2(R.x + V.y, V.x - R.y);
            }
        }
    }
};

// water_level_decon (forward.f90:447-470) on the deposited array of the 8-wave kernels, in place: slot(k) holds the
// numerator, slot(n - k) the denominator of bin k (side[0..1]: the denominators of the DC and Nyquist bins); leaves
// Z = (num conj(den) / max(|den|^2, wlvl)) flt, Hermitian-extended.  Ends with a barrier.
__device__ __forceinline__ void w8_water_level(const DeviceTables &t, double2 *a, const double2 *side, double *red, int itrc,
                                               int tid)
{
    constexpr int n = 4096, nh = 2049;
    const int wave = tid >> 6, lane = tid & 63;
    double m = -HUGE_VAL;
    for (int k = tid; k < nh; k += W8_THREADS) {
        const bool self = (k == 0 || k == 2048);
        const double2 x = self ? side[k == 0 ? 0 : 1] : a[w8_pad(w8_pos(n - k))];
        m = fmax(m, x.x * x.x + x.y * x.y);                   // forward.f90:458
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    double mm = red[0];
#pragma unroll
    for (int w = 1; w < W8_THREADS / 64; ++w) mm = fmax(mm, red[w]);
    const double wlvl = 0.001 * mm;                           // forward.f90:460, pcnt = 0.001 (:149)
    const double *__restrict__ flt = t.flt + (size_t)itrc * nh;
    for (int k = tid; k < nh; k += W8_THREADS) {
        const bool self = (k == 0 || k == 2048);
        const int pk = w8_pad(w8_pos(k));
        const int pnk = self ? pk : w8_pad(w8_pos(n - k));
        const double2 y = a[pk];
        const double2 x = self ? side[k == 0 ? 0 : 1] : a[pnk];
        const double amp = x.x * x.x + x.y * x.y;
        const double dd = fmax(amp, wlvl);                    // forward.f90:464
        const double2 yx = cmul(y, make_double2(x.x, -x.y));
        const double fk = flt[k];
        const double2 R = make_double2(yx.x / dd * fk, yx.y / dd * fk);
        if (self) {
            a[pk] = make_double2(R.x, 0.0);
        } else {
            a[pk] = R;
            a[pnk] = make_double2(R.x, -R.y);
        }
    }
    __syncthreads();
}

// Inverse FFT (radix-8 passes of stride 1, 8, 64 through LDS, the last -- stride 512 -- in registers), vertical
// maximum, shift / normalise / store and misfit of one trace whose Z sits in `a`: the tail of the 8-wave kernels.
// The tail is a chain of latencies (nothing else of the block runs meanwhile, and in the common-ray kernel nothing of
// the CU either), so: the twiddles of a pass (L2-resident table) are requested while the PREVIOUS pass still does its
// butterflies and LDS writes and waits at its barrier -- they have arrived when the pass starts; the vertical maximum
// crosses the waves by one LDS atomic (ds_max_f64) and ONE barrier instead of a store / barrier / load round with two;
// the integer shift (a call with a division) is formed before the transform, not between the maximum and the stores.
template <int SL, bool LEAN = false>   // the twiddles of the pass of stride 2^SL (SL = 3, 6, 9): w[k] = exp(+2 pi i jp k 2^(9 - SL) / 4096)
__device__ __forceinline__ void w8_twiddles(double2 (&w)[8], const double2 *__restrict__ tw, int tid)
{
    const int jp = tid & ((1 << SL) - 1);
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (!LEAN || k == 1 || k == 2 || k == 4) w[k] = tw[(unsigned)((jp * k) << (9 - SL))];   // (< 4096: the table holds the full turn)
}

// one in-place radix-8 pass with this pass's twiddles already in w; afterwards w holds the NEXT pass's (NEXT = 0: none)
// LEAN (the common-ray kernel, which keeps 32 VGPRs of spectra alive across the transform): only w[1], w[2], w[4]
// are loaded; w3 = w1 w2, w5 = w4 w1, w6 = w4 w2, w7 = w4 w3 are formed as they are used (each within 2 ulp of the
// table's value) -- 12 instead of 28 registers of twiddles next to the 32 of the butterfly.
template <int SL, int NEXT, bool LEAN = false>
__device__ __forceinline__ void w8_pass_pf(double2 *a, const double2 *__restrict__ tw, int tid, double2 (&w)[8])
{
    constexpr int STRIDE = 1 << SL;
    const int jp = tid & (STRIDE - 1);
    const int base = ((tid >> SL) << (SL + 3)) + jp;
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a[w8_pad(base + (k << SL))];
    if (SL > 0 && LEAN) {
        const double2 w3 = cmul(w[1], w[2]);
        v[1] = cmul(v[1], w[1]);
        v[2] = cmul(v[2], w[2]);
        v[3] = cmul(v[3], w3);
        v[7] = cmul(v[7], cmul(w[4], w3));
        v[5] = cmul(v[5], cmul(w[4], w[1]));
        v[6] = cmul(v[6], cmul(w[4], w[2]));
        v[4] = cmul(v[4], w[4]);
    } else if (SL > 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], w[k]);
    }
    if (NEXT > 0) w8_twiddles<NEXT, LEAN>(w, tw, tid);
    dft_regs<3>(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[w8_pad(base + (bitrev_small<3>(k) << SL))] = v[k];
}

// AHEAD: request a pass's twiddles one pass ahead (fused8_kernel).  The common-ray kernel, which keeps a walker's
// spectra in 32 VGPRs across this function, loads them inside the pass instead: the 28 registers of twiddles in
// flight across a barrier do not fit its 128-register budget (measured: the spills cost what the prefetch saves).
template <bool AHEAD, bool LEAN = !AHEAD>
__device__ __forceinline__ void w8_fft_store(const TraceParams &P, double2 *a, double *mis, double *red, int ib, int itrc,
                                             int walker, int ipha, bool decon, double tp, int slot, int tid)
{
    const DeviceTables &t = P.t;
    constexpr int n = 4096;
    const int nsmp = t.nsmp;
    const int lane = tid & 63;
    const double2 *__restrict__ tw = t.twiddle;
    const int npre = calc_npre(t.t_start, tp, t.delta, ipha);
    if (tid == 0) red[0] = -HUGE_VAL;      // the vertical maximum's cell (three barriers ahead of its first use)
    double2 w[8];
    if (AHEAD) w8_twiddles<3, LEAN>(w, tw, tid);
    w8_pass_pf<0, 0>(a, tw, tid, w);
    __syncthreads();
    if (!AHEAD) w8_twiddles<3, LEAN>(w, tw, tid);
    w8_pass_pf<3, AHEAD ? 6 : 0, LEAN>(a, tw, tid, w);
    __syncthreads();
    if (!AHEAD) w8_twiddles<6, LEAN>(w, tw, tid);
    w8_pass_pf<6, AHEAD ? 9 : 0, LEAN>(a, tw, tid, w);
    __syncthreads();
    if (!AHEAD) w8_twiddles<9, LEAN>(w, tw, tid);
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a[w8_pad(tid + (k << 9))];
    if (LEAN) {
        const double2 w3 = cmul(w[1], w[2]);
        v[1] = cmul(v[1], w[1]);
        v[2] = cmul(v[2], w[2]);
        v[3] = cmul(v[3], w3);
        v[7] = cmul(v[7], cmul(w[4], w3));
        v[5] = cmul(v[5], cmul(w[4], w[1]));
        v[6] = cmul(v[6], cmul(w[4], w[2]));
        v[4] = cmul(v[4], w[4]);
    } else {
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], w[k]);
    }
    dft_regs<3>(v);                        // v[k] = sample tid + (bitrev3(k) << 9): .x RF trace, .y vertical trace
    RFGPU_ABLATE_AT(2, );
    double fac = 1.0;
    if (!decon) {
        double m = -HUGE_VAL;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmax(m, v[k].y);
        m = wave_max(m);
        // (NaN: ds_max_f64 and fmax both return the other operand, like the store / load round this replaces)
        if (lane == 0) (void)__hip_atomic_fetch_max(&red[0], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        fac = red[0];                                                // maxval(rx) forward.f90:201
    }
    double *__restrict__ dst = P.w.rft + (((size_t)slot * P.w.nslots + walker) * t.ntrc + itrc) * (size_t)P.w.trace_len;
    const double *__restrict__ obs = t.obs + (size_t)itrc * nsmp;
    double *__restrict__ mis_g = P.w.misfit + ((size_t)ib * t.ntrc + itrc) * t.mis_stride;   // defer mode only
    double *xout = (P.extra_out && ib == 0) ? P.extra_out + (size_t)itrc * n : nullptr;
    // (the same store loop as tail_in_registers: one signed reciprocal per thread, 0-based masked sample index)
    const double rfac = (ipha == 1 ? 1.0 : -1.0) / fac;
    const int at0 = ipha == 1 ? tid + npre : npre - tid - 1;        // sample j = tid + 1
    const int step = ipha == 1 ? 512 : -512;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned i0 = (unsigned)(at0 + step * bitrev_small<3>(k)) & 4095u;      // 0-based sample of rft
        const double val = v[k].x * rfac;                            // forward.f90:202 (see rfac)
        if (i0 < (unsigned)P.w.trace_len) __builtin_nontemporal_store(val, &dst[i0]);
        if (xout) xout[i0] = val;
        if (i0 < (unsigned)nsmp) {
            const double m = val - obs[i0];                          // likelihood.f90:88
            // (both stores, not an either / or: hipcc turns `defer ? mis_g : mis` into ONE store through a select of a
            // global and an LDS pointer -- a flat store -- and its backend then fails on the common-ray kernel
            // ("Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base", ROCm 7.2); the LDS copy is one
            // ds_write for the first nsmp samples)
            if (P.defer_logl != 2) mis[i0] = m;      // (2: long windows, no misfit rows in LDS)
            if (P.defer_logl) mis_g[i0] = m;
        }
    }
}

size_t fused8_lds_bytes(int nsmp, int nlay_pad)
{
    // FFT array | misfits | reductions | side | (generic path only) layer constants
    // (side: [0..1] decon denominators of the DC / Nyquist bins, [2..3] fusedc_kernel's Nyquist pair)
    return sizeof(double2) * (size_t)(w8_pad(4095) + 2) + sizeof(double) * (size_t)(((nsmp + 1) & ~1) + 8) +
           sizeof(double2) * 4 + spectra_lds_bytes(nlay_pad);
}

template <int NCOL>
__global__ __launch_bounds__(W8_THREADS, 4) void fused8_kernel(FusedParams F)
{
    extern __shared__ double2 lds2[];
    const TraceParams &P = F.tp;
    const DeviceTables &t = P.t;
    constexpr int n = 4096, nh = 2049;
    double2 *a = lds2;
    double *mis = reinterpret_cast<double *>(a + ((w8_pad(4095) + 2) & ~1));
    double *red = mis + ((t.lds_nsmp + 1) & ~1);
    double2 *side = reinterpret_cast<double2 *>(red + 8);
    double *coef = reinterpret_cast<double *>(side + 4);          // generic path only
    double *tail = coef + (size_t)P.b.nlay_pad * NCOEF;

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    RFGPU_STAGGER(bid);
    if (F.order_next && bid == P.b.nb * t.ntrc) {
        // the dispatch order of the next launch (see fused_kernel)
        int *w = reinterpret_cast<int *>(lds2);
        order_block(P.b.nb, P.b.nlay, P.b.fwd_flag, F.order_next, w, w + 256, w + 512);
        return;
    }
    const int itrc = bid % t.ntrc;            // == forward-trace index here (nfwd == ntrc)
    const int ib = P.b.order ? P.b.order[bid / t.ntrc] : bid / t.ntrc;
    if (bid == 0 && tid == 0) *P.slow_count = 0;
    if (item_not_evaluated(P.b, P.w, t, ib, itrc == 0 && tid == 0)) return;
    const int walker = P.b.walker_ids[ib];
    const int ipha = t.ipha[itrc];
    const bool decon = t.deconv_mode == 1;

    const int bfi = ib * t.ntrc + itrc;
    const int nl = P.b.nlay[ib];
    const int stage_flags = P.w.gflag[bfi];
    const bool sea = stage_flags & 1;             // beta(1) < 0  (forward.f90:229)
    const int ilay0 = sea ? 1 : 0;
    const bool generic = (stage_flags & 2) != 0 || sea != (NCOL == 3);
    const KPtr gcoef = as_scalar_ptr(P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF);
    const KPtr gtail = as_scalar_ptr(P.w.gtail + (size_t)bfi * GTAIL);
    const int slot = 1 - P.w.cur_slot[walker];

    // ---- propagator phase: 8 waves x one 4-bin chain each -> Z in LDS ------------------------------------
    const int nh_eff = (t.nh_active && !decon) ? t.nh_active[itrc] : nh;
    SpectraParams sp = F.sp;
    sp.t.nh = nh_eff;
    sp.nsplit = W8_THREADS / 64;
    for (int k = nh_eff + tid; k < nh; k += W8_THREADS) {
        a[w8_pad(w8_pos(k))] = make_double2(0.0, 0.0);
        if (k != 0 && k != 2048) a[w8_pad(w8_pos(n - k))] = make_double2(0.0, 0.0);
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int lane_m = (64 - lane) & 63;
    const W8Sink sink{a, side, t.flt + (size_t)itrc * nh, nh_eff, ipha, decon,
                      ((lane & 7) << 9) + ((lane >> 3) << 6), ((lane_m & 7) << 9) + ((lane_m >> 3) << 6)};
    if (generic) {
        int nl2, il2;
        bool sea2;
        (void)load_staged(P.w, P.b, t.ntrc, ib, itrc, coef, tail, nl2, il2, sea2);
        if (sea)
            spectra_body<0, 3, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, wave, lane);
        else
            spectra_body<0, 2, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, sink, wave, lane);
    } else {
        // chains start from the block's anchor table (spectra_chunk_chain) when the walker's layers fit it
        const int nsolid = nl - 1 - ilay0;
        const int cap = ((w8_pad(4095) + 2) & ~1) / (2 * ANCHOR_ROW);
        if (nsolid >= 1 && nsolid <= cap && nh_eff == nh) {
            build_anchor_table<W8_THREADS>(a, P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF, nsolid, ilay0, W8_THREADS / 64,
                                           64 * 4, tid);
            __syncthreads();
            spectra_body<4, NCOL, true, W8Sink, KPtr, true, true>(sp, gcoef, gtail, nl, ilay0, ipha, sink, wave, lane, a);
        } else {
            spectra_body<4, NCOL, true, W8Sink, KPtr, true>(sp, gcoef, gtail, nl, ilay0, ipha, sink, wave, lane);
        }
    }
    __syncthreads();
    RFGPU_ABLATE_AT(1, );
    const double tp = decon ? 0.0 : gtail[17];

    if (decon) w8_water_level(t, a, side, red, itrc, tid);
    w8_fft_store<true, true>(P, a, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tid);
    RFGPU_ABLATE_AT(2, );
    RFGPU_ABLATE_AT(3, );
    if (P.defer_logl) return;   // quadratic form and logL: phi_deferred_kernel
    __syncthreads();
    // ---- phi = (misfit . R^-1) . misfit (likelihood.f90:92-93) and logL, as in trace_tail ------------------
    const double phi = quad_form(t, itrc, mis, reinterpret_cast<double *>(a), red, tid);
    if (tid == 0) {
        double *phis = P.w.phi + ((size_t)slot * P.w.nslots + walker) * t.ntrc;
        bool last = true;
        if (t.ntrc > 1) {
            __hip_atomic_store(phis + itrc, phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = atomicAdd(P.w.done + ib, 1) == t.ntrc - 1;
            if (last) P.w.done[ib] = 0;
        } else {
            phis[itrc] = phi;
        }
        if (last) {
            P.b.logl[ib] = logl_from_phi(phis, P.b.sig + (size_t)ib * t.ntrc, t.ntrc, t.nsmp, t.ntrc > 1);
            P.w.prop_fwd[walker] = 1;
        }
    }
}

// ---------------------------------------------------------------------------
// fusedc_kernel: "single FWD mode" (common ray geometry, forward.f90:59-91,141) in ONE launch for nfft = 4096 on land:
// every trace has the same ray parameter and phase, so ONE propagator pass per walker feeds all ntrc traces, which
// differ only by their Gaussian filter flt(:, itrc) (:168,198).  One 512-thread block per WALKER:
//   1. the propagator phase of fused8_kernel (8 waves x one 4-bin chain from the block's anchor table), but the
//      finished bins stay where they are -- each lane keeps the (ur, uz) pairs of its four bins in REGISTERS (32
//      VGPRs; the Nyquist bin's pair, one lane of the block, in LDS: NyqSink) -- instead of going to the FFT array;
//   2. for itrc = 1 .. ntrc: deposit Z = RF flt_t + i V flt_t from those registers (the very code of fused8_kernel's
//      sink: a common-ray trace is bit-identical to the single-trace answer of its own filter), water level if
//      deconv_mode = 1, inverse FFT, vertical maximum, shift / normalise / store, misfit (w8_fft_store), quadratic form;
//   3. logL by the block itself (all traces of a walker are its own: no cross-block hand-off).
// The spectra never travel through HBM: the split plan (spectra_kernel -> spec[nb][2][nh] complex128 -> trace_kernel,
// re-read once per trace) moved 3.9x the algorithmic bytes at the C4 shape and ran its trace kernel at 8 % of the
// FP64 peak.  LDS: the FFT array only (70.7 KB) -> two blocks per CU, four waves per SIMD: one block's latency-bound
// tails overlap the other's propagator phase.  Walkers on the generic path (rare) lay their spectra out as a plain
// table in the FFT array's space first and pick their bins up from there.
// ---------------------------------------------------------------------------
struct NyqSink {
    double2 *nyq;                      // LDS [2]: (ur, uz) of the Nyquist bin, the one bin outside the waves' chains
    __device__ __forceinline__ double weight(int) const { return 1.0; }
    __device__ __forceinline__ void operator()(int k, double2 r, double2 z, double, int) const
    {
        if (k == 2048) {
            nyq[0] = r;
            nyq[1] = z;
        }
    }
};

struct TableSink {
    double2 *tab;         // [2][2049]: ur, then uz, by bin
    __device__ __forceinline__ double weight(int) const { return 1.0; }
    __device__ __forceinline__ void operator()(int k, double2 r, double2 z, double, int) const
    {
        if (k < 2049) {
            tab[k] = r;
            tab[2049 + k] = z;
        }
    }
};

template <int NCOL>
__global__ __launch_bounds__(W8_THREADS, 4) void fusedc_kernel(FusedParams F)
{
    extern __shared__ double2 lds2[];
    const TraceParams &P = F.tp;
    const DeviceTables &t = P.t;
    constexpr int nh = 2049;
    const int nsmp = t.nsmp, ntrc = t.ntrc;
    double2 *a = lds2;
    double *mis = reinterpret_cast<double *>(a + ((w8_pad(4095) + 2) & ~1));
    double *red = mis + ((t.lds_nsmp + 1) & ~1);
    double2 *side = reinterpret_cast<double2 *>(red + 8);
    double *coef = reinterpret_cast<double *>(side + 4);          // generic path only
    double *tail = coef + (size_t)P.b.nlay_pad * NCOEF;

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    RFGPU_STAGGER(bid);
    if (F.order_next && bid == P.b.nb) {
        // the dispatch order of the next launch (see fused_kernel)
        int *w = reinterpret_cast<int *>(lds2);
        order_block(P.b.nb, P.b.nlay, P.b.fwd_flag, F.order_next, w, w + 256, w + 512);
        return;
    }
    const int ib = P.b.order ? P.b.order[bid] : bid;
    if (bid == 0 && tid == 0) *P.slow_count = 0;
    if (item_not_evaluated(P.b, P.w, t, ib, tid == 0)) return;
    const int walker = P.b.walker_ids[ib];
    const int ipha = t.ipha[0];                   // common to every trace (check_ray, forward.f90:59-91)
    const bool decon = t.deconv_mode == 1;

    const int bfi = ib;                           // one forward computation per walker (nfwd = 1)
    const int nl = P.b.nlay[ib];
    const int stage_flags = P.w.gflag[bfi];
    const bool sea = stage_flags & 1;             // beta(1) < 0  (forward.f90:229)
    const int ilay0 = sea ? 1 : 0;
    const bool generic = (stage_flags & 2) != 0 || sea != (NCOL == 3);
    const KPtr gcoef = as_scalar_ptr(P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF);
    const KPtr gtail = as_scalar_ptr(P.w.gtail + (size_t)bfi * GTAIL);
    const int slot = 1 - P.w.cur_slot[walker];

    // ---- 1. propagator phase: the lane's four bins 64 (4 wave + m) + lane end up in ur / uz ---------------------
    SpectraParams sp = F.sp;
    sp.nsplit = W8_THREADS / 64;
    const int wave = tid >> 6, lane = tid & 63;
    double2 ur[4], uz[4];
    if (generic) {
        int nl2, il2;
        bool sea2;
        (void)load_staged(P.w, P.b, 1, ib, 0, coef, tail, nl2, il2, sea2);
        const TableSink ts{a};
        if (sea)
            spectra_body<0, 3, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, ts, wave, lane);
        else
            spectra_body<0, 2, false>(sp, (const double *)coef, (const double *)tail, nl, ilay0, ipha, ts, wave, lane);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = 64 * (4 * wave + m) + lane;
            ur[m] = a[k];
            uz[m] = a[nh + k];
        }
        if (tid == 0) {
            side[2] = a[2048];
            side[3] = a[nh + 2048];
        }
    } else {
        const NyqSink rs{side + 2};
        const int nsolid = nl - 1 - ilay0;
        const int cap = ((w8_pad(4095) + 2) & ~1) / (2 * ANCHOR_ROW);
        if (nsolid >= 1 && nsolid <= cap) {
            build_anchor_table<W8_THREADS>(a, P.w.gcoef + (size_t)bfi * P.b.nlay_pad * NCOEF, nsolid, ilay0, W8_THREADS / 64,
                                           64 * 4, tid);
            __syncthreads();
            spectra_body<4, NCOL, true, NyqSink, KPtr, true, true>(sp, gcoef, gtail, nl, ilay0, ipha, rs, wave, lane, a, ur, uz);
        } else {
            spectra_body<4, NCOL, true, NyqSink, KPtr, true>(sp, gcoef, gtail, nl, ilay0, ipha, rs, wave, lane, nullptr, ur, uz);
        }
    }
    RFGPU_ABLATE_AT(1, );
    // (tp, the direct-arrival time, is common to the traces too: forward.f90:141 skips its recomputation)
    const double tp = decon ? 0.0 : gtail[17];
    double *phis = P.w.phi + ((size_t)slot * P.w.nslots + walker) * ntrc;

    // ---- 2. one trace after the other from the same spectra ---------------------------------------------------------
    for (int itrc = 0; itrc < ntrc; ++itrc) {
        __syncthreads();   // the array is free: table / anchor reads, the previous trace's last FFT pass and quadratic form are done
        // The thread index goes through an empty asm in every iteration: everything derived from it below -- the ~70 LDS
        // and global addresses of the deposits, the four FFT passes and the stores -- is then recomputed per trace
        // (a few integer instructions each) instead of being hoisted out of the loop and kept alive across it, which
        // at the kernel's 128-VGPR budget pushed the spectra and half of those addresses into scratch (242 spilled
        // VGPRs, 6.4 GB of scratch traffic per C4-shaped launch)
        int tix = tid;
        asm volatile("" : "+v"(tix));
        const int wv = tix >> 6, ln = tix & 63, ln_m = (64 - ln) & 63;
        const W8Sink sink{a, side, t.flt + (size_t)itrc * nh, nh, ipha, decon, ((ln & 7) << 9) + ((ln >> 3) << 6),
                          ((ln_m & 7) << 9) + ((ln_m >> 3) << 6)};
        double wgt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) wgt[m] = sink.weight(64 * (4 * wv + m) + ln);
#pragma unroll
        for (int m = 0; m < 4; ++m) sink(64 * (4 * wv + m) + ln, ur[m], uz[m], wgt[m]);
        if (tix == 0) sink(2048, side[2], side[3], sink.weight(2048));
        __syncthreads();
        if (decon) w8_water_level(t, a, side, red, itrc, tix);
        w8_fft_store<false>(P, a, mis, red, ib, itrc, walker, ipha, decon, tp, slot, tix);
        RFGPU_ABLATE_DO(2, continue);     // timing split: no quadratic form
        RFGPU_ABLATE_DO(3, continue);
        RFGPU_ABLATE_AT(6, );             // ... one trace only
        if (!P.defer_logl) {
            // phi = (misfit . R^-1) . misfit (likelihood.f90:92-93); the array doubles as quad_form's scratch
            __syncthreads();
            const double phi = quad_form(t, itrc, mis, reinterpret_cast<double *>(a), red, tid);
            if (tid == 0) phis[itrc] = phi;
        }
    }
    if (P.defer_logl) return;   // quadratic forms and logL: phi_deferred_kernel
    if (tid == 0) {
        P.b.logl[ib] = logl_from_phi(phis, P.b.sig + (size_t)ib * ntrc, ntrc, nsmp, false);
        P.w.prop_fwd[walker] = 1;
    }
}

void launch_fusedc(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int *slow_count, int ablate,
                   int defer_logl, int *order_next, double *extra_out, hipStream_t s)
{
    FusedParams F{};
    F.order_next = order_next;
    F.sp = SpectraParams{t, b, nullptr, W8_THREADS / 64, nullptr, slow_count, w};
    F.tp = TraceParams{t, b, nullptr, w, 12, {}, slow_count, ablate, defer_logl, extra_out};
    const size_t lds = fused8_lds_bytes(t.lds_nsmp, b.nlay_pad);
    const dim3 grid((unsigned)b.nb + (order_next ? 1u : 0u));
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(fusedc_kernel<2>));
    hipLaunchKernelGGL((fusedc_kernel<2>), grid, dim3(W8_THREADS), lds, s, F);
}

void launch_fused8(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int *slow_count, int ablate,
                   int defer_logl, int *order_next, double *extra_out, hipStream_t s)
{
    FusedParams F{};
    F.order_next = order_next;
    F.sp = SpectraParams{t, b, nullptr, W8_THREADS / 64, nullptr, slow_count, w};
    F.tp = TraceParams{t, b, nullptr, w, 12, {}, slow_count, ablate, defer_logl, extra_out};
    const size_t lds = fused8_lds_bytes(t.lds_nsmp, b.nlay_pad);
    const dim3 grid((unsigned)(b.nb * t.ntrc) + (order_next ? 1u : 0u));
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(fused8_kernel<2>));
    hipLaunchKernelGGL((fused8_kernel<2>), grid, dim3(W8_THREADS), lds, s, F);
}

template <int BK, int NCOL>
static void launch_fused_one(dim3 grid, size_t lds, hipStream_t s, const FusedParams &F)
{
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(fused_kernel<BK, NCOL>));
    hipLaunchKernelGGL((fused_kernel<BK, NCOL>), grid, dim3(TRACE_THREADS), lds, s, F);
}

template <int NCOL>
static void launch_fused_ncol(int chain, dim3 grid, size_t lds, hipStream_t s, const FusedParams &F)
{
    switch (chain) {
    case 2: launch_fused_one<2, NCOL>(grid, lds, s, F); break;
    case 3: launch_fused_one<3, NCOL>(grid, lds, s, F); break;
    case 4: launch_fused_one<4, NCOL>(grid, lds, s, F); break;
    case 8: launch_fused_one<8, NCOL>(grid, lds, s, F); break;
    default: launch_fused_one<0, NCOL>(grid, lds, s, F); break;
    }
}

void launch_fused(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int chain, int *slow_count,
                  int ablate, int defer_logl, int *order_next, double *extra_out, hipStream_t s)
{
    FusedParams F{};
    F.order_next = order_next;
    F.sp = SpectraParams{t, b, nullptr, TRACE_THREADS / 64, nullptr, slow_count, w};
    F.tp = TraceParams{t, b, nullptr, w, 0, {}, slow_count, ablate, defer_logl, extra_out};
    while ((1 << F.tp.log2n) < t.nfft) ++F.tp.log2n;
    F.tp.plan = make_fft_plan(F.tp.log2n);
    const size_t lds = fused_lds_bytes(t.nfft, t.lds_nsmp, b.nlay_pad);
    const dim3 grid((unsigned)(b.nb * t.ntrc) + (order_next ? 1u : 0u));
    if (t.sdep > 0.0)
        launch_fused_ncol<3>(chain, grid, lds, s, F);
    else
        launch_fused_ncol<2>(chain, grid, lds, s, F);
}

// likelihood.f90:81-93 for a trace supplied by the host (the fwd_flag = .false. branch
// of the single-call drop-in: the Fortran host owns rft(:,:,chain)): one block per trace
// reads the first nsmp samples of the scratch walker's proposal slot.
__global__ __launch_bounds__(TRACE_THREADS) void phi_kernel(DeviceTables t, WalkerState w, int walker)
{
    extern __shared__ double2 lds2[];
    const int nsmp = t.nsmp;
    double *work = reinterpret_cast<double *>(lds2);                  // [4 * nsmp]
    double *mis = work + 4 * (size_t)((nsmp + 1) & ~1);
    double *red = mis + ((nsmp + 1) & ~1);
    const int tid = threadIdx.x, itrc = blockIdx.x;
    const int slot = 1 - w.cur_slot[walker];
    const double *src = w.rft + (((size_t)slot * w.nslots + walker) * t.ntrc + itrc) * (size_t)w.trace_len;
    const double *obs = t.obs + (size_t)itrc * nsmp;
    for (int i = tid; i < nsmp; i += TRACE_THREADS) mis[i] = src[i] - obs[i];   // likelihood.f90:88
    __syncthreads();
    const double phi = quad_form(t, itrc, mis, work, red, tid);
    if (tid == 0) w.phi[((size_t)slot * w.nslots + walker) * t.ntrc + itrc] = phi;
}

void launch_phi(const DeviceTables &t, const WalkerState &w, int walker, hipStream_t s)
{
    const size_t lds = sizeof(double) * (size_t)(5 * ((t.nsmp + 1) & ~1) + 8);   // 80 KB at the reference's npts_max = 2000
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(phi_kernel));
    hipLaunchKernelGGL(phi_kernel, dim3((unsigned)t.ntrc), dim3(TRACE_THREADS), lds, s, t, w, walker);
}

size_t trace_lds_bytes(int nfft, int nsmp, int nlay_pad)
{
    return sizeof(double) * (trace_work_doubles(nfft, nsmp, nlay_pad) + (size_t)(((nsmp + 1) & ~1) + 8));
}

void launch_trace(const DeviceTables &t, const BatchArgs &b, const double2 *spec, const WalkerState &w,
                  int *slow_count, double2 *xbuf, int defer_logl, hipStream_t s)
{
    TraceParams P{t, b, spec, w, 0, {}, slow_count, 0, defer_logl, nullptr, xbuf};
    if (t.twiddle_any) {   // nfft is not a power of two: direct-DFT variants
        if (xbuf) {
            static LdsOptIn opt_big;
            opt_big(reinterpret_cast<const void *>(trace_anyn_kernel<true>));
            hipLaunchKernelGGL(trace_anyn_kernel<true>, dim3((unsigned)(b.nb * t.ntrc)), dim3(TRACE_THREADS),
                               trace_anyn_big_lds_bytes(t.nfft, t.lds_nsmp), s, P);
        } else {
            static LdsOptIn opt_any;
            opt_any(reinterpret_cast<const void *>(trace_anyn_kernel<false>));
            hipLaunchKernelGGL(trace_anyn_kernel<false>, dim3((unsigned)(b.nb * t.ntrc)), dim3(TRACE_THREADS),
                               trace_anyn_lds_bytes(t.nfft, t.lds_nsmp, b.nlay_pad), s, P);
        }
        return;
    }
    while ((1 << P.log2n) < t.nfft) ++P.log2n;
    P.plan = make_fft_plan(P.log2n);
    const size_t lds = trace_lds_bytes(t.nfft, t.lds_nsmp, b.nlay_pad);
    static LdsOptIn opt;
    opt(reinterpret_cast<const void *>(trace_kernel));
    hipLaunchKernelGGL(trace_kernel, dim3((unsigned)(b.nb * t.ntrc)), dim3(TRACE_THREADS), lds, s, P);
}

// ---------------------------------------------------------------------------
// log-likelihood from cached quadratic forms (likelihood.f90:86,94-96), one thread per item;
// used for host-owned traces (rf_calc_likelihood_of_trace) -- the batched path forms logL in
// trace_tail or, for large batches, in the follow-up kernels below (defer_logl)
// ---------------------------------------------------------------------------
struct LoglParams {
    DeviceTables t;
    BatchArgs b;
    WalkerState w;
};

__global__ __launch_bounds__(256) void logl_kernel(LoglParams P)
{
#pragma clang fp contract(off)
    const int ib = blockIdx.x * blockDim.x + threadIdx.x;
    if (ib >= P.b.nb) return;
    const int walker = P.b.walker_ids[ib];
    const int fwd = P.b.fwd_flag ? P.b.fwd_flag[ib] : 1;
    const int cur = P.w.cur_slot[walker];
    // fwd_flag = .false. re-uses the stored trace (likelihood.f90:81): same phi
    const int slot = fwd ? 1 - cur : cur;
    const double *phi = P.w.phi + ((size_t)slot * P.w.nslots + walker) * P.t.ntrc;
    double ll = 0.0;
    for (int it = 0; it < P.t.ntrc; ++it) {
        const double sg = P.b.sig[(size_t)ib * P.t.ntrc + it];
        const double q = 0.5 * phi[it] / (sg * sg);
        const double r = (double)P.t.nsmp * log(sg);
        ll = ll - q - r;
    }
    P.b.logl[ib] = ll;
    P.w.prop_fwd[walker] = fwd;
}

// Follow-up kernel of a batch whose fused kernel ran with defer_logl (likelihood.f90:87-98): the quadratic forms of
// every trace and logL, ONE launch.
//
// phi_deferred_kernel: one 256-thread block per (PHI_W batch items, trace).  The arithmetic is quad_form's, operation
// for operation (lanes own columns j of R^-1, the four waves own contiguous row quarters, quarter sums combined in
// wave order, the final dot product by a wave reduction) -- so the values are bit-identical to the in-kernel path --
// but a row of R^-1 is fetched once for PHI_W items instead of once per block.  The block that finishes a group's
// LAST trace forms the group's logL (likelihood.f90:94-96, same operation order, no FMA contraction): the quadratic
// forms cross blocks by agent-scope stores, a drained write and a counter, like the in-kernel multi-trace hand-off.
// (Until round 3 a second kernel, one thread per item, formed logL: one launch and ~5 us of stream time more per
// batch.  Measured alternatives, all slower than its 55 us at the C4 shape: one block walking a group's traces in
// turn, 56 us; that with 512 threads, one column per lane and 32 row loads in flight, 60 us; the misfits as scalar
// operands from an item-interleaved layout instead of LDS, 78 us.  With the row loads grouped (PHI_ROWS): 51.5 us.
// The block is a chain of latencies -- misfits in, rows of R^-1 from L2, partial sums across the waves, the
// hand-off's drained store and counter -- with 4 blocks per CU to hide them behind.)
__global__ __launch_bounds__(256) void phi_deferred_kernel(LoglParams P)
{
    extern __shared__ double lds[];
    const int nsmp = P.t.nsmp, ntrc = P.t.ntrc;
    double *mis = lds;                                    // [PHI_W][nsmp]
    double *part = mis + (size_t)PHI_W * nsmp;            // [4][PHI_W][nsmp]
    double *red = part + (size_t)4 * PHI_W * nsmp;        // [4][PHI_W]
    int *flag = reinterpret_cast<int *>(red + 4 * PHI_W); // [1] "this block finishes the group"
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int it = blockIdx.x % ntrc, grp = blockIdx.x / ntrc, ib0 = grp * PHI_W;
    for (int e = tid; e < PHI_W * nsmp; e += 256) {
        const int w = e / nsmp, i = e - w * nsmp, ib = ib0 + w;
        const bool live = ib < P.b.nb && P.w.item_state[ib] == 1;
        mis[e] = live ? P.w.misfit[((size_t)ib * ntrc + it) * P.t.mis_stride + i] : 0.0;
    }
    __syncthreads();
    const double *__restrict__ RT = P.t.r_inv_t + (size_t)it * nsmp * nsmp;
    const int rows = (nsmp + 3) >> 2;
    const int r0 = wv * rows, r1 = min(nsmp, r0 + rows);
    // two columns per lane and pass (j, j + 64): twice the loads in flight per row
    for (int j = lane; j < nsmp; j += 128) {
        const int j2 = j + 64;
        const bool two = j2 < nsmp;
        double acc0[PHI_W], acc1[PHI_W];
#pragma unroll
        for (int w = 0; w < PHI_W; ++w) acc0[w] = acc1[w] = 0.0;
        // rows in groups of PHI_ROWS: every load of a group is in flight before its first use (the loop is a chain of
        // L2 latencies otherwise: 7 of them for nsmp = 101 at four rows per trip); rows stay in ascending order
        for (int g0 = r0; g0 < r1; g0 += PHI_ROWS) {
            double xa[PHI_ROWS], xb[PHI_ROWS];
#pragma unroll
            for (int u = 0; u < PHI_ROWS; ++u) {
                const int i = g0 + u < r1 ? g0 + u : r1 - 1;
                xa[u] = RT[(size_t)i * nsmp + j];
                xb[u] = two ? RT[(size_t)i * nsmp + j2] : 0.0;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < PHI_ROWS; ++u) {
                if (g0 + u < r1) {
#pragma unroll
                    for (int w = 0; w < PHI_W; ++w) {
                        const double m = mis[w * nsmp + g0 + u];
                        acc0[w] = fma(m, xa[u], acc0[w]);
                        acc1[w] = fma(m, xb[u], acc1[w]);
                    }
                }
            }
        }
#pragma unroll
        for (int w = 0; w < PHI_W; ++w) {
            part[((size_t)wv * PHI_W + w) * nsmp + j] = acc0[w];
            if (two) part[((size_t)wv * PHI_W + w) * nsmp + j2] = acc1[w];
        }
    }
    __syncthreads();
    double acc[PHI_W];
#pragma unroll
    for (int w = 0; w < PHI_W; ++w) acc[w] = 0.0;
    for (int j = tid; j < nsmp; j += 256) {
#pragma unroll
        for (int w = 0; w < PHI_W; ++w) {
            const double *pw = part + (size_t)w * nsmp + j;
            const size_t q = (size_t)PHI_W * nsmp;
            const double phi1 = ((pw[0] + pw[q]) + pw[2 * q]) + pw[3 * q];
            acc[w] = fma(phi1, mis[w * nsmp + j], acc[w]);
        }
    }
#pragma unroll
    for (int w = 0; w < PHI_W; ++w) {
        const double v = wave_sum(acc[w]);
        if (lane == 0) red[wv * PHI_W + w] = v;
    }
    __syncthreads();
    const int ib = ib0 + tid;
    const bool mine = tid < PHI_W && ib < P.b.nb && P.w.item_state[ib] == 1;
    const int walker = mine ? P.b.walker_ids[ib] : 0;
    double *phis = P.w.phi + ((size_t)(mine ? 1 - P.w.cur_slot[walker] : 0) * P.w.nslots + walker) * ntrc;
    const double phi = (red[tid & (PHI_W - 1)] + red[PHI_W + (tid & (PHI_W - 1))]) +
                       (red[2 * PHI_W + (tid & (PHI_W - 1))] + red[3 * PHI_W + (tid & (PHI_W - 1))]);
    if (ntrc == 1) {
        // nothing to wait for: logL right here
        if (mine) {
            phis[0] = phi;
            P.b.logl[ib] = logl_from_phi(&phi, P.b.sig + ib, 1, nsmp, false);
            P.w.prop_fwd[walker] = 1;
        }
        return;
    }
    // hand-off between the ntrc blocks of a group (see trace_tail): write-through stores of the quadratic forms,
    // drained, then the group's counter (done[] of the group's first item; self-resetting); the last arriver reads
    // every trace's value with agent-scope loads
    if (mine) __hip_atomic_store(phis + it, phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const bool last = atomicAdd(P.w.done + ib0, 1) == ntrc - 1;
        if (last) P.w.done[ib0] = 0;
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    if (*flag && mine) {
        P.b.logl[ib] = logl_from_phi(phis, P.b.sig + (size_t)ib * ntrc, ntrc, nsmp, true);
        P.w.prop_fwd[walker] = 1;
    }
}

size_t phi_deferred_lds_bytes(int nsmp) { return sizeof(double) * ((size_t)5 * PHI_W * nsmp + 4 * PHI_W + 2); }

void launch_logl_deferred(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s)
{
    LoglParams P{t, b, w};
    const unsigned groups = (unsigned)((b.nb + PHI_W - 1) / PHI_W);
    hipLaunchKernelGGL(phi_deferred_kernel, dim3(groups * (unsigned)t.ntrc), dim3(256), phi_deferred_lds_bytes(t.nsmp), s, P);
}

void launch_logl(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s)
{
    LoglParams P{t, b, w};
    hipLaunchKernelGGL(logl_kernel, dim3((unsigned)((b.nb + 255) / 256)), dim3(256), 0, s, P);
}

// ---------------------------------------------------------------------------
// Long time windows: phi_gemm_kernel.  The reference evaluates phi1 = matmul(misfits, r_inv) for any window up to
// npts_max = 2000 samples (src/likelihood.f90:92-93, src/params.f90:44); r_inv is dense, so the work grows with
// nsmp^2 (1201 samples, a 60 s window: 11.5 MB of R^-1 per trace, 2.9 Mflop per (walker, trace)).  quad_form re-reads
// the whole matrix per block and phi_deferred_kernel once per 8 walkers -- a matrix-vector product at 0.25 flop per
// byte.  Beyond what phi_deferred_kernel's LDS holds (nsmp > 191) the quadratic forms of a batch are instead ONE GEMM
//     Phi1[nb x nsmp] = M_t[nb x nsmp] . R^-1_t[nsmp x nsmp]      per trace t, then  phi = rowsum(Phi1 o M_t)
// on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), the only dense contraction of the path.
// R^-1 is NOT assumed symmetric (it is only to ~1e-10).  By default the product runs on the quadratic
// form's UPPER TRIANGLE T (triangle_r_inv, rfgpu_api.cpp: T(i, j) = R^-1(i, j) + R^-1(j, i) above the diagonal, the
// diagonal, 0 below): m R m^T = sum_j m_j sum_{i <= j} m_i T(i, j) holds for any matrix -- only its symmetric part
// enters the form -- and column block c then needs rows 0 .. 64 (c + 1) - 1 only: half the multiply-adds, half the
// matrix traffic (c4w60: GEMM + logL 1.17 -> 0.68 ms, profiles/EXPERIMENTS.md).  The rounding differs from the reference's
// row-vector x matrix product the way any other summation order does (checked against the oracle at every window in
// the tests).  "gemm_triangle" = 0 runs the full product on R^-1 itself, row index ascending as in the reference.
//
// Tiling: a 256-thread block owns 128 walkers x 128 columns; its four waves 64 x 64 each (4 x 4 MFMA tiles, 16
// accumulators of 4 doubles per lane).  K advances in steps of 16 through LDS: the A tile (128 misfit rows x 16) and
// the B tile (16 rows of R^-1 x 128) are fetched once per block and step -- R^-1 once per 128 walkers instead of once
// per 1 .. 8 -- with the next step's global loads in flight behind the current step's 64 MFMAs per wave; two blocks per
// CU cover each other's barriers.  Operand layout of the instruction (lane l): A[row l & 15][k = l >> 4],
// B[k = l >> 4][col l & 15], D[row (l >> 4) + 4 r][col l & 15] in register r = 0 .. 3.
// Both operands are zero-padded in HBM (misfit rows to kp = nsmp rounded up to 16, the R^-1 image to kp x np, np =
// nsmp rounded up to 128), so the loop has no edge cases; sub-tiles wholly outside the batch or the window are skipped.
// Epilogue: each lane multiplies its accumulators by the matching misfits and adds them up over its four column tiles
// (ascending), a 16-lane butterfly finishes the row's sum over the wave's 64 columns, and the per-chunk partial goes to
// part[trace][chunk][item]; phi_gemm_finish_kernel adds the chunks in ascending order and forms logL
// (likelihood.f90:94-96).  Every step has a fixed order that does not depend on the batch: a chain evaluated alone or
// in a full batch gets bit-identical values.
// ---------------------------------------------------------------------------
constexpr int PG_BK = 16;
constexpr int PG_LDA = PG_BK + 1;      // doubles per LDS row of the A tile ([row][k]; odd: the fragment reads spread over the banks)

struct PhiGemmParams {
    const double *mis;     // [>= nb][ntrc][ld] misfits, rows zero-padded to ld = kp
    const double *rg;      // [ntrc][kp][np]: R^-1, or its quadratic form's upper triangle (tri)
    double *part;          // [ntrc][nchunk][pstride]
    int nb, ntrc, ld, kp, np, nchunk, pstride;
    int tri;               // rg is upper triangular: column block n0 needs rows 0 .. n0 + BN - 1 only
};

typedef double pg_acc_t __attribute__((ext_vector_type(4)));
typedef double pg_v2_t __attribute__((ext_vector_type(2)));

// Two tilings of the same product (the values do not depend on the choice: every output element is the same k-ordered
// chain of MFMAs, every 64-column partial the same sums):
//   WN = 1 (default): block 128 x 64, waves 4 (M) x 1, wave tile 32 x 64 (MT = 2 row tiles): 123 VGPRs, 27 KB of LDS,
//           FOUR blocks per CU -- four waves per SIMD keep the matrix pipe fed across each other's barriers and staging,
//           small launches spread evenly (a 20 s window on 8192 walkers is 8 GFLOP), and a window's last column block
//           carries at most 48 columns of padding.  Measured (tests/tools/gemm_tile_ab.py, profiles/r04_gemm_tile_ab.txt):
//           full product: c4w60 1.17 ms against 1.34 ms for the wide tile, c4w20 0.19 against 0.21; triangle: 0.67 against 0.78.
//   WN = 2 ("gemm_tile" = 128): block 128 x 128, waves 2 x 2, wave tile 64 x 64 (MT = 4): 16 flop per byte staged
//           instead of 10.7, 208 VGPRs, two blocks per CU.  The first version; kept behind the option.
// (A third variant on v_mfma_f64_4x4x4_4b_f64 -- its four blocks as one 4 x 16 x 4 product, same bits -- was measured
// slower, LDS-bound: profiles/EXPERIMENTS.md; tools/mfma_f64_4x4_layout.hip has the operand layout.)
template <int WN>
__global__ __launch_bounds__(256, WN == 2 ? 2 : 4) void phi_gemm_kernel(PhiGemmParams G)
{
    constexpr int BM = 128, BN = 64 * WN, WM = 4 / WN, MT = BM / WM / 16;   // MT row tiles of 16 per wave
    constexpr int LDB = BN + 16;       // doubles per LDS row of the B tile ([k][col]; rows 16 doubles apart modulo the banks)
    constexpr int NA = BM / 32, NB = BN / 32;                               // double2 per thread and K step: A tile, B tile
    __shared__ double As[BM * PG_LDA];
    __shared__ __attribute__((aligned(16))) double Bs[PG_BK * LDB];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    // blocks of one (trace, column block) are consecutive: they share the B tile (L2) and stream their own A tiles
    const int nmb = (G.nb + BM - 1) / BM, nnb = (G.kp + BN - 1) / BN;
    int bid = blockIdx.x;
    const int mblk = bid % nmb;
    bid /= nmb;
    // (triangular image: the column blocks with the long K loops first, the short ones fill in behind them)
    const int nblk = G.tri ? nnb - 1 - bid % nnb : bid % nnb, it = bid / nnb;
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int wm = wv % WM, wn = wv / WM;
    const int mw = m0 + 16 * MT * wm, nw = n0 + 64 * wn;
    // live 16-row / 16-column sub-tiles of this wave (wave-uniform)
    const int mt_live = min(MT, max(0, (G.nb - mw + 15) >> 4));
    const int nt_live = min(4, max(0, (G.kp - nw) >> 4));
    const bool full = mt_live == MT && nt_live == 4;

    const size_t a_rs = (size_t)G.ntrc * G.ld;                                   // doubles between the rows of two items
    const double *__restrict__ Ag = G.mis + (size_t)it * G.ld;
    const double *__restrict__ Bg = G.rg + (size_t)it * G.kp * G.np + n0;
    // staging: A tile BM rows x 8 double2, B tile 16 rows x BN / 2 double2.  Rows beyond the batch re-read its last
    // row: their accumulators are never written anywhere.  (np, a multiple of 128, covers every column block.)
    const pg_v2_t *ap[NA], *bp[NB];
    pg_v2_t pa[NA], pb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + 256 * i;
        const int row = min(m0 + (e >> 3), G.nb - 1);
        ap[i] = reinterpret_cast<const pg_v2_t *>(Ag + (size_t)row * a_rs + 2 * (e & 7));
        pa[i] = *ap[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + 256 * i;
        bp[i] = reinterpret_cast<const pg_v2_t *>(Bg + (size_t)(e / (BN / 2)) * G.np + 2 * (e % (BN / 2)));
        pb[i] = *bp[i];
    }
    const size_t b_step = (size_t)(PG_BK / 2) * G.np;      // double2 per K step of the B image
    pg_acc_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = pg_acc_t{0.0, 0.0, 0.0, 0.0};

    // Triangular image: rows beyond the block's last column are zero.  (A 128-wide block runs its left 64 columns
    // through 64 rows of zeros: acc + m * 0 leaves every accumulator as it is, so both tilings still give the same bits.)
    const int nkt = (G.tri ? min(G.kp, n0 + BN) : G.kp) / PG_BK;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                  // the previous step's fragments are read
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            double *da = As + (e >> 3) * PG_LDA + 2 * (e & 7);
            da[0] = pa[i].x;
            da[1] = pa[i].y;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<pg_v2_t *>(Bs + (e / (BN / 2)) * LDB + 2 * (e % (BN / 2))) = pb[i];
        }
        __syncthreads();
        if (kt + 1 < nkt) {                               // the next step's tiles: in flight behind this step's MFMAs
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ap[i] += PG_BK / 2;
                pa[i] = *ap[i];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                bp[i] += b_step;
                pb[i] = *bp[i];
            }
        }
#pragma unroll
        for (int kk = 0; kk < PG_BK / 4; ++kk) {
            double av[MT], bv[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = As[(16 * MT * wm + 16 * mt + l15) * PG_LDA + 4 * kk + l4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = Bs[(4 * kk + l4) * LDB + 64 * wn + 16 * nt + l15];
            if (full) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (mt < mt_live && nt < nt_live)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    // ---- phi partial of the wave's 64 columns for each of its rows -----------------------------------------------------
    double *__restrict__ pout = G.part + ((size_t)it * G.nchunk + (nw >> 6)) * G.pstride;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mw + 16 * mt + l4 + 4 * r;
            const bool live = row < G.nb;
            const double *__restrict__ mrow = Ag + (size_t)(live ? row : 0) * a_rs + nw + l15;
            double sum = 0.0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (nt < nt_live) sum = fma(acc[mt][nt][r], live ? mrow[16 * nt] : 0.0, sum);
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64);
            if (l15 == 0 && live) pout[row] = sum;
        }
    }
}

// phi = the chunk partials in ascending order, then logL (likelihood.f90:94-96): one thread per batch item.  Items
// that did not run the forward model were finished by the trace kernels themselves (cached phi / NaN).
__global__ __launch_bounds__(256) void phi_gemm_finish_kernel(LoglParams P, const double *__restrict__ part, int nlive,
                                                              int nchunk, int pstride)
{
#pragma clang fp contract(off)
    const int ib = blockIdx.x * blockDim.x + threadIdx.x;
    if (ib >= P.b.nb) return;
    if (P.w.item_state[ib] != 1) return;
    const int ntrc = P.t.ntrc, walker = P.b.walker_ids[ib];
    double *phis = P.w.phi + ((size_t)(1 - P.w.cur_slot[walker]) * P.w.nslots + walker) * ntrc;
    for (int it = 0; it < ntrc; ++it) {
        const double *q = part + (size_t)it * nchunk * pstride + ib;
        double s = 0.0;
        for (int c = 0; c < nlive; ++c) s = s + q[(size_t)c * pstride];
        phis[it] = s;
    }
    P.b.logl[ib] = logl_from_phi(phis, P.b.sig + (size_t)ib * ntrc, ntrc, P.t.nsmp, false);
    P.w.prop_fwd[walker] = 1;
}

void launch_phi_gemm(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, const PhiGemmTables &g, hipStream_t s)
{
    PhiGemmParams G{w.misfit, g.triangle ? g.rt : g.rg, g.part, b.nb, t.ntrc, t.mis_stride, g.kp, g.np, g.nchunk, g.pstride, g.triangle};
    // the tiling (same values either way: see the kernel): 128 x 64 blocks unless the option asks for 128 x 128
    const unsigned nmb = (unsigned)((b.nb + 127) / 128);
    const unsigned nnb2 = (unsigned)((g.kp + 127) / 128), nnb1 = (unsigned)((g.kp + 63) / 64);
    if (g.tile == 128)
        hipLaunchKernelGGL(phi_gemm_kernel<2>, dim3(nmb * nnb2 * (unsigned)t.ntrc), dim3(256), 0, s, G);
    else
        hipLaunchKernelGGL(phi_gemm_kernel<1>, dim3(nmb * nnb1 * (unsigned)t.ntrc), dim3(256), 0, s, G);
    LoglParams P{t, b, w};
    hipLaunchKernelGGL(phi_gemm_finish_kernel, dim3((unsigned)((b.nb + 255) / 256)), dim3(256), 0, s, P, g.part,
                       (g.kp + 63) / 64, g.nchunk, g.pstride);
}

// likelihood.f90:88 for a trace supplied by the host, long windows: the misfits of the scratch walker's proposal slot
// go to the misfit row of batch item 0 (phi_gemm_kernel with nb = 1 follows)
__global__ __launch_bounds__(256) void misfit_of_trace_kernel(DeviceTables t, WalkerState w, int walker)
{
    const int itrc = blockIdx.x;
    const int slot = 1 - w.cur_slot[walker];
    const double *src = w.rft + (((size_t)slot * w.nslots + walker) * t.ntrc + itrc) * (size_t)w.trace_len;
    const double *obs = t.obs + (size_t)itrc * t.nsmp;
    double *dst = w.misfit + (size_t)itrc * t.mis_stride;
    for (int i = threadIdx.x; i < t.nsmp; i += 256) dst[i] = src[i] - obs[i];
    if (itrc == 0 && threadIdx.x == 0) w.item_state[0] = 1;   // (no stage_kernel on this path: the one item is to be finished)
}

void launch_misfit_of_trace(const DeviceTables &t, const WalkerState &w, int walker, hipStream_t s)
{
    hipLaunchKernelGGL(misfit_of_trace_kernel, dim3((unsigned)t.ntrc), dim3(256), 0, s, t, w, walker);
}

// ---------------------------------------------------------------------------
// format_model on the device (reference src/model.f90:175-290 with vp_to_rho :298-314 and
// the 3-array quick_sort of src/sort.f90:34-68): one wave per walker turns
// (k, z, dVp, dVs) into the layer stack (alpha, beta, rho, h).  All of it is bookkeeping
// that must be bit-exact -- the same unstable quicksort (ties keep the reference's
// permutation), nint() table look-ups, no FMA contraction, single-precision literals.
// ---------------------------------------------------------------------------
__device__ __noinline__ double vp_to_rho_dev(double a1)
{
#pragma clang fp contract(off)
    const double a2 = a1 * a1, a3 = a2 * a1, a4 = a3 * a1, a5 = a4 * a1;
    return (double)1.6612f * a1 - (double)0.4721f * a2 + (double)0.0671f * a3 - (double)0.0043f * a4 +
           (double)0.000106f * a5;
}

__device__ __forceinline__ void swap3(double *a, double *b, double *c, int i, int j)
{
    double t = a[i]; a[i] = a[j]; a[j] = t;
    t = b[i]; b[i] = b[j]; b[j] = t;
    t = c[i]; c[i] = c[j]; c[j] = t;
}

// the reference's recursive quick_sort (1-based il..ir inclusive) with an explicit stack
__device__ void quick_sort3_dev(double *a, double *b, double *c, int n)
{
    int stack_l[32], stack_r[32], sp = 0;
    stack_l[0] = 1;
    stack_r[0] = n;
    sp = 1;
    while (sp > 0) {
        --sp;
        const int il = stack_l[sp], ir = stack_r[sp];
        if (ir - il <= 0) continue;
        const int ipiv = (il + ir) / 2;
        const double piv = a[ipiv - 1];
        swap3(a, b, c, ipiv - 1, ir - 1);
        int i = il;
        for (int j = il; j <= ir; ++j)
            if (a[j - 1] < piv) {
                swap3(a, b, c, i - 1, j - 1);
                ++i;
            }
        swap3(a, b, c, i - 1, ir - 1);
        // the reference recurses into (il, i) and then (i + 1, ir); the two sub-ranges are disjoint,
        // so the order in which they are processed does not change the result.  Pushing the larger
        // one first (the smaller is popped next) bounds the stack by log2(n) + 1 entries.
        const int ll = il, lr = i, rl = i + 1, rr = ir;
        if (lr - ll >= rr - rl) {
            stack_l[sp] = ll; stack_r[sp] = lr; ++sp;
            stack_l[sp] = rl; stack_r[sp] = rr; ++sp;
        } else {
            stack_l[sp] = rl; stack_r[sp] = rr; ++sp;
            stack_l[sp] = ll; stack_r[sp] = lr; ++sp;
        }
    }
}

__device__ __forceinline__ int f_nint_dev(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : -floor(0.5 - x)); }

// One WAVE per walker (until round 4: one thread, whose quicksort walked three arrays in global scratch -- 350 us for any
// batch at k_max 30, a third of the C4 evaluation's stream time once the sampler's proposals went down unformatted).
// The interfaces sit in LDS; lane i ranks interface i by counting the shallower ones and scatters it to its place:
// with distinct depths the sorted order is unique, so this IS the reference quicksort's result.  Equal depths -- where
// the unstable quicksort's permutation decides which perturbation belongs to which layer -- are detected (a wave
// vote) and such a walker is sorted by lane 0 with the reference's algorithm itself.  Then one lane per layer: table
// look-up, velocities, density, thickness, the validity rules; the verdict is a wave vote.
constexpr int FM_WAVES = 4;     // walkers per block

__global__ __launch_bounds__(64 * FM_WAVES) void format_model_kernel(FormatParams P)
{
#pragma clang fp contract(off)
    extern __shared__ double fm_lds[];             // [FM_WAVES][6][kmax]: z, dVp, dVs as given | sorted
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ib = blockIdx.x * FM_WAVES + wv;
    if (ib >= P.nb) return;                        // (wave-uniform: no barrier below, waves work alone)
    const ModelConfig &m = P.m;
    const int kmax = m.k_max, k = P.k[ib];
    const int fwd = P.fwd_in ? P.fwd_in[ib] : 1;
    if (fwd != 1) {          // sigma-only (or skipped) item: no model to format
        if (lane == 0) {
            P.nlay[ib] = 2;
            P.flag[ib] = fwd;
            if (P.valid) P.valid[ib] = 1;
        }
        return;
    }
    if (k < 1 || k >= kmax) {   // (the host entry points refuse such a batch; a device array may hold anything)
        if (lane == 0) {
            P.nlay[ib] = 2;
            P.flag[ib] = -1;
            if (P.valid) P.valid[ib] = 0;
        }
        return;
    }
    double *uz = fm_lds + (size_t)wv * 6 * kmax, *uvp = uz + kmax, *uvs = uvp + kmax;
    double *tz = uvs + kmax, *tvp = tz + kmax, *tvs = tvp + kmax;
    const int ldz = P.ldz > 0 ? P.ldz : kmax - 1;   // (the batched Fortran host keeps z(k_max, nchains))
    for (int i = lane; i < kmax; i += 64) {
        uz[i] = i < kmax - 1 ? P.z[(size_t)ib * ldz + i] : 0.0;
        uvp[i] = P.dvp[(size_t)ib * kmax + i];
        uvs[i] = P.dvs[(size_t)ib * kmax + i];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- sort the first k interfaces by depth (:197-198) ----------------------------------------------------
    bool tie = false;
    for (int i = lane; i < k; i += 64) {
        const double zi = uz[i];
        int rank = 0;
        for (int j2 = 0; j2 < k; ++j2) {
            const double zj = uz[j2];
            rank += zj < zi;
            tie |= (zj == zi) & (j2 != i);
        }
        tz[rank] = zi;
        tvp[rank] = uvp[i];
        tvs[rank] = uvs[i];
    }
    if (__any(tie)) {
        // equal depths: the reference's quicksort itself, on the arrays as given (lane 0)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            for (int i = 0; i < k; ++i) { tz[i] = uz[i]; tvp[i] = uvp[i]; tvs[i] = uvs[i]; }
            quick_sort3_dev(tz, tvp, tvs, k);
        }
    }
    // (entries beyond k keep their places: only the half-space's perturbations, index kmax - 1, are read)
    for (int i = lane; i < kmax; i += 64)
        if (i >= k) { tvp[i] = uvp[i]; tvs[i] = uvs[i]; }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- one lane per layer ------------------------------------------------------------------------------------
    const int pad = P.nlay_pad;
    double *A = P.layers + (size_t)ib * 4 * pad, *B = A + pad, *R = B + pad, *H = R + pad;
    const int i0 = m.sdep > 0.0 ? 1 : 0;
    if (i0 && lane == 0) {                                                  // :201-207
        A[0] = 1.5; B[0] = -999.0; R[0] = 1.0; H[0] = m.sdep;
    }
    bool ok = true;
    for (int j = 1 + lane; j <= k + 1; j += 64) {
        // j = 1: top layer (:210-231), 2..k: middle layers (:235-262), k+1: half-space (:265-282)
        double zc, thick;
        int idx;
        if (j == 1) {
            zc = 0.5 * (m.sdep + tz[0]);
            thick = tz[0] - m.sdep;
            idx = 0;
        } else if (j <= k) {
            zc = 0.5 * (tz[j - 1] + tz[j - 2]);
            thick = tz[j - 1] - tz[j - 2];
            idx = j - 1;
        } else {
            zc = 0.5 * (m.z_max + tz[k - 1]);
            thick = 999.0;
            idx = kmax - 1;
        }
        const int iz = f_nint_dev((zc - m.z_ref_min) / m.dz_ref) + 1;
        const double b = m.vs_ref[iz - 1] + tvs[idx];
        const double a = m.vp_mode == 1 ? m.vp_ref[iz - 1] + tvp[idx] : m.vp_ref[iz - 1];
        if (a < m.vp_min || a > m.vp_max || b < m.vs_min || b > m.vs_max || a / b < m.vpvs_min ||
            a / b > m.vpvs_max)
            ok = false;
        const int i = i0 + j - 1;
        A[i] = a; B[i] = b; R[i] = vp_to_rho_dev(a); H[i] = thick;
        if (j == 1 && thick < (double)0.125f * a) ok = false;                // :229 (not h_min)
        if (j > 1 && j <= k && thick < m.h_min) ok = false;                  // :256
    }
    ok = __all(ok);
    if (lane == 0) {
        P.nlay[ib] = i0 + k + 1;
        P.flag[ib] = ok ? 1 : -1;
        if (P.valid) P.valid[ib] = ok ? 1 : 0;
    }
}

void launch_format_model(const FormatParams &P, hipStream_t s)
{
    const size_t lds = sizeof(double) * FM_WAVES * 6 * (size_t)P.m.k_max;
    hipLaunchKernelGGL(format_model_kernel, dim3((unsigned)((P.nb + FM_WAVES - 1) / FM_WAVES)), dim3(64 * FM_WAVES), lds, s, P);
}

// ---------------------------------------------------------------------------
// Longest-processing-time-first dispatch order.  A block's cost is proportional to the
// walker's layer count (2 .. k_max); blocks are handed to CUs in index order, so with few
// rounds of blocks per CU (C2: 1024 blocks on 512 slots) a deep walker dispatched late sets
// the kernel time.  A counting sort by descending layer count (one block, LDS histogram)
// gives order[], and block b works on batch item order[b / ntrc].  Values do not depend on it.
// ---------------------------------------------------------------------------
// the counting sort, by one block of >= 256 threads; hist / start: 256 ints of LDS each, wave_tot: 4
__device__ __forceinline__ void order_block(int nb, const int *nlay, const int *fwd_flag, int *order, int *hist,
                                            int *start, int *wave_tot)
{
    const int tid = threadIdx.x;   // key = min(nlay, 255); items without a forward model: key 0
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nb; i += blockDim.x) {
        const int key = (fwd_flag && fwd_flag[i] != 1) ? 0 : max(0, min(nlay[i], 255));
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    // exclusive prefix sum over descending keys: thread t < 256 owns key 255 - t; a shuffle scan inside
    // each of the four waves, then the totals of the waves before it
    int incl = 0, mine = 0;
    if (tid < 256) {
        mine = hist[255 - tid];
        incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        int before = 0;
        for (int w = 0; w < (tid >> 6); ++w) before += wave_tot[w];
        start[255 - tid] = before + incl - mine;
    }
    __syncthreads();
    for (int i = tid; i < nb; i += blockDim.x) {
        const int key = (fwd_flag && fwd_flag[i] != 1) ? 0 : max(0, min(nlay[i], 255));
        order[atomicAdd(&start[key], 1)] = i;
    }
}

__global__ __launch_bounds__(1024) void order_kernel(int nb, const int *nlay, const int *fwd_flag, int *order)
{
    __shared__ int hist[256], start[256], wave_tot[4];
    order_block(nb, nlay, fwd_flag, order, hist, start, wave_tot);
}

void launch_order(int nb, const int *nlay, const int *fwd_flag, int *order, hipStream_t s)
{
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, s, nb, nlay, fwd_flag, order);
}

// ---------------------------------------------------------------------------
// gather the first nout samples of many walkers' traces into one packed buffer
// out[n][ntrc][nout] (for the posterior amplitude histogram, pt_mcmc.f90:272-285)
// ---------------------------------------------------------------------------
__global__ void gather_rft_kernel(WalkerState w, int ntrc, int nfft, int n, const int *walker_ids, int which,
                                  int nout, double *out)
{
    const int i = blockIdx.x / ntrc, itrc = blockIdx.x % ntrc;
    if (i >= n) return;
    const int wk = walker_ids[i];
    const int cur = w.cur_slot[wk];
    const int slot = (which == 1 && w.prop_fwd[wk]) ? 1 - cur : cur;
    const double *src = w.rft + (((size_t)slot * w.nslots + wk) * ntrc + itrc) * (size_t)w.trace_len;
    double *dst = out + ((size_t)i * ntrc + itrc) * (size_t)nout;
    for (int j = threadIdx.x; j < nout; j += blockDim.x) dst[j] = src[j];
}

void launch_gather_rft(const WalkerState &w, int ntrc, int nfft, int n, const int *walker_ids, int which, int nout,
                       double *out, hipStream_t s)
{
    hipLaunchKernelGGL(gather_rft_kernel, dim3((unsigned)(n * ntrc)), dim3(128), 0, s, w, ntrc, nfft, n, walker_ids,
                       which, nout, out);
}

// ---------------------------------------------------------------------------
// accept step (pt_mcmc.f90:190): flip the walker's current-trace slot
// ---------------------------------------------------------------------------
__global__ void commit_kernel(WalkerState w, int nb, const int *walker_ids, const int *accept)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int wk = walker_ids[i];
    if (wk < 0 || wk >= w.nslots) {            // (rf_commit_device cannot check its ids on the host: refuse, report)
        if (w.err[0] == 0) {
            w.err[1] = i;
            w.err[2] = wk;
            w.err[0] = 2;
        }
        return;
    }
    if (accept[i] && w.prop_fwd[wk]) w.cur_slot[wk] = 1 - w.cur_slot[wk];
    w.prop_fwd[wk] = 0;
}

void launch_commit(const WalkerState &w, int nb, const int *walker_ids, const int *accept, int ntrc,
                   hipStream_t s)
{
    (void)ntrc;
    hipLaunchKernelGGL(commit_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, w, nb,
                       walker_ids, accept);
}

// ---------------------------------------------------------------------------
// judge_pt (pt_mcmc.f90:580-595) for a list of DISJOINT pairs, one thread per pair
// (the reference proposes one pair per iteration; the batched schedule draws a
// random partial matching, so no walker appears twice and pairs commute).
// ---------------------------------------------------------------------------
__global__ void pt_swap_kernel(int npairs, const int *pairs, const double *log_u, double *temps,
                               const double *logl, int *accepted)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const int c1 = pairs[2 * i], c2 = pairs[2 * i + 1];
    const double t1 = temps[c1], t2 = temps[c2];
    const double del_s = (logl[c2] - logl[c1]) * (1.0 / t1 - 1.0 / t2);
    const int yn = log_u[i] <= del_s;
    if (yn) {
        temps[c2] = t1;
        temps[c1] = t2;
    }
    if (accepted) accepted[i] = yn;
}

void launch_pt_swap(int npairs, const int *pairs, const double *log_u, double *temps, const double *logl,
                    int *accepted, hipStream_t s)
{
    hipLaunchKernelGGL(pt_swap_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, npairs, pairs,
                       log_u, temps, logl, accepted);
}

// The same decisions on the GATHERED ensemble of a multi-rank run (pt_mcmc.f90:508-511: global id = rank * nchains +
// chain): g_temps / g_logl [nranks * nchains] are every rank's (T, logL) as the all-gather delivered them -- rank
// blocks in rank order, i.e. already indexed by global id -- and are only read; a swap writes the temperature this
// rank's own chain holds afterwards straight into `temps` [nchains] (temperatures move, states stay: :532-535).
// Pairs are disjoint, so every decision sees the pre-swap snapshot exactly as the in-place kernel does.
__global__ void pt_swap_gathered_kernel(int npairs, const int *pairs, const double *log_u, const double *g_temps,
                                        const double *g_logl, int nchains, int rank, int nranks, double *temps, int *accepted)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const int c1 = pairs[2 * i], c2 = pairs[2 * i + 1];
    // (a pair outside the gathered ensemble -- a schedule drawn for another nchains -- is no pair: nothing is read or moved)
    const int n_all = nranks * nchains;
    if (c1 < 0 || c2 < 0 || c1 >= n_all || c2 >= n_all) {
        if (accepted) accepted[i] = 0;
        return;
    }
    const double t1 = g_temps[c1], t2 = g_temps[c2];
    const double del_s = (g_logl[c2] - g_logl[c1]) * (1.0 / t1 - 1.0 / t2);
    const int yn = log_u[i] <= del_s;
    if (yn) {
        const int lo = rank * nchains;
        if (c2 >= lo && c2 < lo + nchains) temps[c2 - lo] = t1;
        if (c1 >= lo && c1 < lo + nchains) temps[c1 - lo] = t2;
    }
    if (accepted) accepted[i] = yn;
}

void launch_pt_swap_gathered(int npairs, const int *pairs, const double *log_u, const double *g_temps,
                             const double *g_logl, int nchains, int rank, int nranks, double *temps, int *accepted,
                             hipStream_t s)
{
    hipLaunchKernelGGL(pt_swap_gathered_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, npairs, pairs,
                       log_u, g_temps, g_logl, nchains, rank, nranks, temps, accepted);
}

} // namespace rfgpu
