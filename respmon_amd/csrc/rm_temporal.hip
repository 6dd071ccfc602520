// respmon_amd/csrc/rm_temporal.hip -- temporal filters (transforms.py:38-102) and the materialised min/max mask (transforms.py:184-192)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// ------------------------------------------------------------------------------------------
// temporal operator  (transforms.py:82-102)
// ------------------------------------------------------------------------------------------
// scipy.fftpack.fftfreq(n, d) == numpy.fft.fftfreq: val = 1.0/(n*d); f[i] = i*val (i < (n-1)/2+1),
// f[i] = (i - n)*val otherwise (i.e. -(n/2) .. -1)
static void band_bounds(int n, double fps, double fmin, double fmax, int *lo, int *hi)
{
    double d = 1.0 / fps;
    double val = 1.0 / (n * d);
    int N = (n - 1) / 2 + 1;
    double best_lo = 0, best_hi = 0;
    *lo = 0; *hi = 0;
    for (int i = 0; i < n; ++i) {
        double f = (double)(i < N ? i : i - n) * val;
        double a = std::fabs(f - fmin), b = std::fabs(f - fmax);
        if (i == 0 || a < best_lo) { best_lo = a; *lo = i; }  // argmin keeps the first minimum
        if (i == 0 || b < best_hi) { best_hi = b; *hi = i; }
    }
}

extern "C" int rm_temporal_operator(int T, double fps, double fmin, double fmax, double *M, int *blo, int *bhi)
{
    if (T < 1 || !(fps > 0) || !M) return fail(RM_E_BADARG, "rm_temporal_operator: bad argument");
    int lo, hi;
    band_bounds(T, fps, fmin, fmax, &lo, &hi);
    if (blo) *blo = lo;
    if (bhi) *bhi = hi;
    const int n = T;
    // keep[k] for the PACKED rfft array: fft[hi:-hi] = 0; if lo != 0: fft[:lo] = 0, fft[-lo:] = 0
    std::vector<char> keep(n, 1);
    {
        // python slice [hi : n-hi] (when hi == 0 the stop is -0 == 0 -> empty slice)
        int start = hi, stop = (hi == 0) ? 0 : n - hi;
        for (int k = start; k < stop; ++k) keep[k] = 0;
        if (lo != 0) {
            for (int k = 0; k < lo && k < n; ++k) keep[k] = 0;
            for (int k = (n - lo > 0 ? n - lo : 0); k < n; ++k) keep[k] = 0;
        }
    }
    // packed real FFT rows: R[0,t] = 1; R[2j-1,t] = cos(2 pi j t / n); R[2j,t] = -sin(2 pi j t / n);
    // (n even) R[n-1,t] = (-1)^t.  Inverse as the reference applies it: Re(ifft(packed))[s] =
    // (1/n) sum_k packed[k] cos(2 pi k s / n).   M[s,t] = (1/n) sum_{k kept} cos(2 pi k s/n) R[k,t]
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<double> R((size_t)n * n, 0.0);
    for (int k = 0; k < n; ++k) {
        if (!keep[k]) continue;
        for (int t = 0; t < n; ++t) {
            double v;
            if (k == 0) v = 1.0;
            else if ((n % 2 == 0) && k == n - 1) v = (t % 2 == 0) ? 1.0 : -1.0;
            else {
                int j = (k + 1) / 2;
                long long jt = ((long long)j * t) % n;  // exact argument reduction
                double ang = two_pi * (double)jt / (double)n;
                v = (k % 2 == 1) ? std::cos(ang) : -std::sin(ang);
            }
            R[(size_t)k * n + t] = v;
        }
    }
    for (int s = 0; s < n; ++s)
        for (int t = 0; t < n; ++t) {
            long double acc = 0.0L;
            for (int k = 0; k < n; ++k) {
                if (!keep[k]) continue;
                long long ks = ((long long)k * s) % n;
                acc += (long double)std::cos(two_pi * (double)ks / (double)n) * (long double)R[(size_t)k * n + t];
            }
            M[(size_t)s * n + t] = (double)(acc / (long double)n);
        }
    return RM_OK;
}

// packed-rfft indices that survive the reference's mask (transforms.py:91-94), in increasing order
static void kept_packed_indices(int n, double fps, double fmin, double fmax, std::vector<int> &kept)
{
    int lo, hi;
    band_bounds(n, fps, fmin, fmax, &lo, &hi);
    std::vector<char> keep(n, 1);
    int start = hi, stop = (hi == 0) ? 0 : n - hi;  // python slice [hi:-hi]; -0 == 0 gives an empty slice
    for (int k = start; k < stop; ++k) keep[k] = 0;
    if (lo != 0) {
        for (int k = 0; k < lo && k < n; ++k) keep[k] = 0;
        for (int k = (n - lo > 0 ? n - lo : 0); k < n; ++k) keep[k] = 0;
    }
    kept.clear();
    for (int k = 0; k < n; ++k)
        if (keep[k]) kept.push_back(k);
}

// Two-stage form of the operator, MERGED and for the UNIQUE output frames (rm_kernels.h sym_frames):
//   packed index k contributes  Re(ifft)[s] += cos(2 pi k s / n) / n * y[k]  (transforms.py:98 on the packed array), and
//   cos(2 pi (n - k) s / n) == cos(2 pi k s / n): the kept indices k and n - k share their inverse column, so their forward rows
//   are added here once and for all.  m = min(k, n - k) names the merged row;
//     Rz[i][t] = R[m_i][t] (if kept) + R[n - m_i][t] (if kept, and a different index)      i < nm, t < n
//     Cz[s][i] = cos(2 pi m_i s / n) / n                                                   s < n / 2 + 1
//   The rows s > n / 2 of the inverse are the mirror images of these (out[n - s] == out[s], as in the reference: scipy's ifft of
//   a real array is exactly Hermitian), so they are never computed.
static double packed_row(int n, int k, int t)
{
    const double two_pi = 6.283185307179586476925286766559;
    if (k == 0) return 1.0;
    if ((n % 2 == 0) && k == n - 1) return (t % 2 == 0) ? 1.0 : -1.0;
    const int j = (k + 1) / 2;
    const long long jt = ((long long)j * t) % n;  // exact argument reduction
    const double ang = two_pi * (double)jt / (double)n;
    return (k % 2 == 1) ? std::cos(ang) : -std::sin(ang);
}

static void merged_operator(int n, const std::vector<int> &kept, std::vector<int> &ms, std::vector<double> &Rz, std::vector<double> &Cz)
{
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<char> is_kept(n, 0);
    for (int k : kept) is_kept[k] = 1;
    ms.clear();
    for (int m = 0; m <= n / 2; ++m) {
        const int k2 = n - m;
        if (is_kept[m] || (m != 0 && k2 < n && is_kept[k2])) ms.push_back(m);
    }
    const int nm = (int)ms.size(), Th = n / 2 + 1;
    Rz.assign((size_t)nm * n, 0.0);
    Cz.assign((size_t)Th * nm, 0.0);
    for (int i = 0; i < nm; ++i) {
        const int m = ms[i], k2 = n - m;
        const bool second = m != 0 && k2 != m && k2 < n && is_kept[k2];
        for (int t = 0; t < n; ++t) {
            double v = is_kept[m] ? packed_row(n, m, t) : 0.0;
            if (second) v = is_kept[m] ? v + packed_row(n, k2, t) : packed_row(n, k2, t);
            Rz[(size_t)i * n + t] = v;
        }
        for (int sidx = 0; sidx < Th; ++sidx) {
            const long long ks = ((long long)m * sidx) % n;
            Cz[(size_t)sidx * nm + i] = std::cos(two_pi * (double)ks / (double)n) / (double)n;
        }
    }
}

// symmetry class of merged row m when n is even: rows 0, odd m (cosine rows, and (-1)^t for k = n - 1) are even in t, even m > 0
// (sine rows) odd in t; the partner n - m has the parity of m, so a merged row never mixes the classes
static bool merged_row_is_even(int m) { return m == 0 || (m & 1); }


int get_operator(rm_ctx *ctx, int T, double fps, double fmin, double fmax, TemporalOp *op, hipStream_t s)
{
    const int Th = T / 2 + 1, nks = (Th + 3) / 4, mt = (Th + 15) / 16;
    if (!(ctx->op_T == T && ctx->op_fps == fps && ctx->op_fmin == fmin && ctx->op_fmax == fmax)) {
        std::vector<int> kept, ms;
        kept_packed_indices(T, fps, fmin, fmax, kept);
        std::vector<double> R, C;
        merged_operator(T, kept, ms, R, C);
        const int nm = (int)ms.size();
        ctx->op_nk = nm;
        double *dR = nullptr, *dC = nullptr;
        RM_TRY(ws(ctx, "temporal_R", R.size() + 1, &dR));
        RM_TRY(ws(ctx, "temporal_C", C.size() + 1, &dC));
        // fragment-major copies for the matrix-core kernel (k_temporal_sym): class-pure tiles of 16 merged rows, NH "even" tiles
        // then NH "odd" ones, zero padded; frames folded to t <= n / 2 (needs an even n)
        std::vector<int> rows_e, rows_o;
        for (int i = 0; i < nm; ++i) (merged_row_is_even(ms[i]) ? rows_e : rows_o).push_back(i);
        const int NH = std::max(1, (int)std::max((rows_e.size() + 15) / 16, (rows_o.size() + 15) / 16));
        const bool mf = nm >= 1 && T % 2 == 0 && T >= 8 && NH <= TM_MAX_HALF;
        const int NT = 2 * NH;
        std::vector<double> Rf(mf ? (size_t)nks * NT * 64 : 1, 0.0), Cf(mf ? (size_t)mt * 4 * NT * 64 : 1, 0.0);
        if (mf) {
            auto row_of = [&](int q, int i) -> int {   // merged row held by row i of tile q, or -1
                const std::vector<int> &v = q < NH ? rows_e : rows_o;
                const size_t j = (size_t)(q < NH ? q : q - NH) * 16 + i;
                return j < v.size() ? v[j] : -1;
            };
            for (int ks = 0; ks < nks; ++ks)
                for (int q = 0; q < NT; ++q)
                    for (int l = 0; l < 64; ++l) {
                        const int r = row_of(q, l & 15), t = 4 * ks + (l >> 4);
                        Rf[((size_t)ks * NT + q) * 64 + l] = (r >= 0 && t < Th) ? R[(size_t)r * T + t] : 0.0;
                    }
            for (int m = 0; m < mt; ++m)
                for (int q = 0; q < NT; ++q)
                    for (int rr = 0; rr < 4; ++rr)
                        for (int l = 0; l < 64; ++l) {
                            const int sI = 16 * m + (l & 15), r = row_of(q, 4 * rr + (l >> 4));
                            Cf[(((size_t)m * NT + q) * 4 + rr) * 64 + l] = (r >= 0 && sI < Th) ? C[(size_t)sI * nm + r] : 0.0;
                        }
        }
        double *dRf = nullptr, *dCf = nullptr;
        RM_TRY(ws(ctx, "temporal_Rf", Rf.size(), &dRf));
        RM_TRY(ws(ctx, "temporal_Cf", Cf.size(), &dCf));
        ctx->op_mfma = mf ? NH : 0;
        if (!R.empty()) {
            HIP_TRY(hipMemcpyAsync(dR, R.data(), sizeof(double) * R.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dC, C.data(), sizeof(double) * C.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dRf, Rf.data(), sizeof(double) * Rf.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dCf, Cf.data(), sizeof(double) * Cf.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(stream_wait(s));  // the sources are stack-lifetime vectors
        }
        ctx->op_T = T; ctx->op_fps = fps; ctx->op_fmin = fmin; ctx->op_fmax = fmax;
    }
    double *dR = nullptr, *dC = nullptr;
    RM_TRY(ws(ctx, "temporal_R", (size_t)ctx->op_nk * T + 1, &dR));
    RM_TRY(ws(ctx, "temporal_C", (size_t)ctx->op_nk * Th + 1, &dC));
    op->R = dR; op->C = dC; op->nk = ctx->op_nk;
    if (ctx->op_mfma) {
        double *dRf = nullptr, *dCf = nullptr;
        RM_TRY(ws(ctx, "temporal_Rf", (size_t)nks * 2 * ctx->op_mfma * 64, &dRf));
        RM_TRY(ws(ctx, "temporal_Cf", (size_t)mt * 8 * ctx->op_mfma * 64, &dCf));
        op->Rf = dRf; op->Cf = dCf; op->tiles = ctx->op_mfma;
    }
    return RM_OK;
}

// out[Th, NP] = amp * Cz (Rz x), x[T, NP]: the Th = T / 2 + 1 unique frames of the band-passed signal (transforms.py:86-99);
// full = true: out is [T, NP] and the mirrored frames are stored as well
int launch_temporal(rm_ctx *ctx, const double *x, int T, size_t NP, const TemporalOp &op, double amp, double *out, hipStream_t s,
                           CollapseState *st_init, bool full)
{
    const int Th = sym_frames(T);
    if (op.nk == 0) {  // nothing survives the mask
        HIP_TRY(hipMemsetAsync(out, 0, sizeof(double) * (size_t)(full ? T : Th) * NP, s));
        if (st_init) { hipLaunchKernelGGL(k_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, st_init); LAUNCH_CHECK(); }
        return RM_OK;
    }
    const int mirror_n = full ? T : 0;
    // large levels: one wave per 16 pixel columns, no K-split (k_temporal_sym_px); the choice depends on (T, NP) only
    int cus_t = 256;
    (void)hipDeviceGetAttribute(&cus_t, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const bool wide = ctx->dbg.temporal_wide >= 0 ? ctx->dbg.temporal_wide != 0 : NP >= (size_t)64 * 4 * cus_t;
    if (op.Rf && !ctx->dbg.temporal_valu && wide) {
        const dim3 grid((unsigned)((NP + 63) / 64)), block(256);
        if (op.tiles == 1) hipLaunchKernelGGL((k_temporal_sym_px<1>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else if (op.tiles == 2) hipLaunchKernelGGL((k_temporal_sym_px<2>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else hipLaunchKernelGGL((k_temporal_sym_px<3>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        LAUNCH_CHECK();
        return RM_OK;
    }
    if (op.Rf && !ctx->dbg.temporal_valu) {
        const dim3 grid((unsigned)((NP + 15) / 16)), block(64 * TM_W);
        if (op.tiles == 1) hipLaunchKernelGGL((k_temporal_sym<1>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else if (op.tiles == 2) hipLaunchKernelGGL((k_temporal_sym<2>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else hipLaunchKernelGGL((k_temporal_sym<3>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        LAUNCH_CHECK();
        return RM_OK;
    }
    const size_t sh1 = sizeof(double) * (size_t)T * TF_KC, sh2 = sizeof(double) * (size_t)op.nk * TF_SC;
    if (sh1 > 64 * 1024 || sh2 > 64 * 1024) return fail(RM_E_UNSUPPORTED, "temporal filter: T=%d exceeds the LDS-staged operator (T <= 2048)", T);
    double *y = nullptr;
    RM_TRY(ws(ctx, "temporal_y", (size_t)op.nk * NP, &y));
    dim3 g1((unsigned)((NP + 63) / 64), (op.nk + TF_KC - 1) / TF_KC), g2((unsigned)((NP + 63) / 64), (Th + TF_SC - 1) / TF_SC);
    hipLaunchKernelGGL(k_temporal_fwd<>, g1, dim3(64), sh1, s, x, T, NP, op.R, op.nk, y, st_init);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_temporal_inv<>, g2, dim3(64), sh2, s, y, op.nk, NP, op.C, Th, amp, out, mirror_n);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_temporal_bandpass_filter_fft(rm_ctx *ctx, const double *data, int T, size_t npix, double fps, double fmin,
                                               double fmax, double amp, double *out, void *stream)
{
    if (!ctx || !data || !out || T < 1 || !(fps > 0)) return fail(RM_E_BADARG, "rm_temporal_bandpass_filter_fft: bad argument");
    if (npix == 0) return RM_OK;
    if (data == out) return fail(RM_E_BADARG, "rm_temporal_bandpass_filter_fft: in-place filtering is not supported");
    hipStream_t s = (hipStream_t)stream;
    TemporalOp op;
    RM_TRY(get_operator(ctx, T, fps, fmin, fmax, &op, s));
    return launch_temporal(ctx, data, T, npix, op, amp, out, s, nullptr, true);
}

extern "C" int rm_time_average(rm_ctx *ctx, const void *data, int dtype, int T, size_t npix, double *out, void *stream)
{
    if (!ctx || !data || !out || T < 1 || !valid_dtype(dtype)) return fail(RM_E_BADARG, "rm_time_average: bad argument");
    if (npix == 0) return RM_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_time_average<uint8_t>), grid, block, 0, s, (const uint8_t *)data, T, npix, out); break;
    case RM_F16: hipLaunchKernelGGL((k_time_average<__half>), grid, block, 0, s, (const __half *)data, T, npix, out); break;
    case RM_F32: hipLaunchKernelGGL((k_time_average<float>), grid, block, 0, s, (const float *)data, T, npix, out); break;
    default: hipLaunchKernelGGL((k_time_average<double>), grid, block, 0, s, (const double *)data, T, npix, out); break;
    }
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_lfilter(rm_ctx *ctx, const double *data, int T, size_t npix, const double *b_host, const double *a_host, int ncoef,
                          double scale, double *out, void *stream)
{
    if (!ctx || !data || !out || !b_host || !a_host || T < 1 || ncoef < 1) return fail(RM_E_BADARG, "rm_lfilter: bad argument");
    if (ncoef > IIR_MAX) return fail(RM_E_UNSUPPORTED, "rm_lfilter: %d coefficients > %d", ncoef, IIR_MAX);
    if (a_host[0] == 0.0) return fail(RM_E_BADARG, "rm_lfilter: a[0] must not be zero");
    if (npix == 0) return RM_OK;
    if (data == out) return fail(RM_E_BADARG, "rm_lfilter: in-place filtering is not supported");
    IirCoef c;
    c.n = ncoef;
    for (int i = 0; i < IIR_MAX; ++i) {  // scipy normalises both polynomials by a[0] first
        c.b[i] = i < ncoef ? b_host[i] / a_host[0] : 0.0;
        c.a[i] = i < ncoef ? a_host[i] / a_host[0] : 0.0;
    }
    hipLaunchKernelGGL(k_lfilter<>, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, (hipStream_t)stream, data, T, npix, c, scale, out);
    LAUNCH_CHECK();
    return RM_OK;
}

// transforms.py:184-192 on a materialised [n] array: min, max, top = max - (max - min) * threshold,
// masked = raw with every value >= top replaced by min
extern "C" int rm_threshold_mask(rm_ctx *ctx, const double *raw, size_t n, double threshold, double *masked, double *minmax_host,
                                 void *stream)
{
    if (!ctx || !raw || n == 0) return fail(RM_E_BADARG, "rm_threshold_mask: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    CollapseState *st = ctx->d_state;
    hipLaunchKernelGGL(k_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, st);
    ctx->state_fresh = false;   // this call reduces into the state
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_plain<>, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, raw, n, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_minmax<>, dim3(1), dim3(NSTRIPE), 0, s, st, threshold);
    LAUNCH_CHECK();
    if (masked) {
        hipLaunchKernelGGL(k_mask_plain<>, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, raw, n, st, masked);
        LAUNCH_CHECK();
    }
    if (minmax_host) {
        HIP_TRY(hipMemcpyAsync(ctx->h_state, st, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
        HIP_TRY(stream_wait(s));
        minmax_host[0] = ctx->h_state->min_val;
        minmax_host[1] = ctx->h_state->max_val;
    }
    return RM_OK;
}

