/*
 * oracle_net.c -- CPU restatement of the Cattus leaf-evaluation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cattus_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it, and there only as the checker / CPU baseline.
 *
 * What it restates (paths relative to the reference checkout):
 *   - planes_to_tensor            engine/src/net/mod.rs:121-156
 *   - ConvNetV1.forward (eval)    training/cattus_train/net_utils.py:4-89
 *   - non-finite logit scrub      engine/src/net/mod.rs:56-61
 *   - calc_moves_probs            engine/src/net/mod.rs:106-119
 *
 * Pinning: tests/test_oracle_golden.py checks this file against fixtures
 * produced by importing the reference's own net_utils.py (script:
 * oracle/gen_golden.py, vectors under tests/golden/) at the reference's
 * cross-runtime tolerance (training/tests/test_net_output.py:28-33).
 *
 * Arithmetic contract ("canonical order").  The reference delegates the
 * arithmetic to third-party runtimes whose summation order is unspecified, so
 * this oracle fixes one, and the HIP f32 path is written to reproduce it bit
 * for bit (v_mfma_f32_32x32x2_f32 is an in-order fmaf chain):
 *   - BatchNorm is folded:  scale = gamma / sqrtf(var + 1e-5f),
 *     w' = w * scale,  b' = beta - mean * scale   (gamma=1, beta=0 if !affine)
 *   - 3x3 conv output = fmaf chain from +0 over
 *       for chunk of 32 input channels, for tap (dy,dx) row-major,
 *       for k in chunk in the order 0,4,1,5,2,6,3,7, 8,12,9,13,...
 *     with zero padding outside the board, then  y = acc + b'
 *     (+ residual input, then ReLU).
 *   - 1x1 conv / linear layers = fmaf chain over k in the same 8-group order
 *     (0,4,1,5,2,6,3,7, 8,12,...), then + bias.
 *   - tanh: oracle_tanhf below (only + - * / fmaf and exponent bit edits).
 * Build with -ffp-contract=off so no other fusion happens.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))
#define HOT __attribute__((target_clones("arch=haswell", "default")))

#define HEADER_BYTES 64
#define BN_EPS 1e-5f
#define MAX_S 11
#define MAX_HW (MAX_S * MAX_S)
#define MAX_PAD ((MAX_S + 2) * (MAX_S + 2))

typedef struct {
    uint32_t C, S, M, blocks, F, vhc, phc, hidden;
} oracle_desc;

typedef struct {
    float *w; /* folded, [tap 9][cout][cin] */
    float *b; /* folded, [cout] */
    uint32_t cin, cout;
} conv3_layer;

typedef struct oracle_net {
    oracle_desc d;
    conv3_layer stem;
    conv3_layer *c1, *c2; /* per residual block */
    float *vconv_w, *vconv_b; /* [vhc][F], [vhc] folded */
    float *pconv_w, *pconv_b; /* [phc][F], [phc] folded */
    const float *vfc1_w, *vfc1_b, *vfc2_w, *vfc2_b, *pfc_w, *pfc_b; /* into blob copy */
    /* SimpleTwoHeadedModel (training/cattus_train/net_utils.py:92-121; blob with F == 0): into the blob copy */
    const float *d1_w, *d1_b, *d2_w, *d2_b, *sv_w, *sv_b, *sp_w, *sp_b;
    float *blob; /* owned copy of tensor payload */
} oracle_net;

/* ---------------------------------------------------------------- tanh --- */

static inline float oracle_expf_pos(float x) {
    /* exp(x) for -80 <= x <= 40 : n = rint(x*log2e), r = x - n*ln2 (two-part), degree-6 poly */
    const float log2e = 1.44269504088896341f;
    const float ln2_hi = 0.693145751953125f;        /* 0x3f317200 */
    const float ln2_lo = 1.42860682030941723e-06f;  /* ln2 - ln2_hi */
    float n = nearbyintf(x * log2e);
    float r = fmaf(n, -ln2_hi, x);
    r = fmaf(n, -ln2_lo, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { float f; uint32_t u; } v;
    v.f = p;
    v.u += ((uint32_t)(int32_t)n) << 23; /* p in [0.7,1.5), n in [-116,58]: stays normal */
    return v.f;
}

ORACLE_API float oracle_tanhf(float x) {
    float ax = fabsf(x);
    float t;
    if (!(ax == ax)) return x; /* NaN */
    if (ax < 0.5f) {
        /* odd Taylor series to x^13; |err| < 3e-8 relative on [0,0.5) */
        float s = ax * ax;
        float p = 21844.0f / 6081075.0f;
        p = fmaf(p, s, -1382.0f / 155925.0f);
        p = fmaf(p, s, 62.0f / 2835.0f);
        p = fmaf(p, s, -17.0f / 315.0f);
        p = fmaf(p, s, 2.0f / 15.0f);
        p = fmaf(p, s, -1.0f / 3.0f);
        t = fmaf(ax * s, p, ax);
    } else if (ax < 10.0f) {
        float e = oracle_expf_pos(2.0f * ax);
        t = 1.0f - 2.0f / (e + 1.0f);
    } else {
        t = 1.0f;
    }
    return x < 0.0f ? -t : t;
}

/* ------------------------------------------------------- plane expand --- */

static inline int plane_bit(const uint64_t *plane, uint32_t idx) {
    return (int)((plane[idx >> 6] >> (idx & 63)) & 1u);
}

/* engine/src/net/mod.rs:121-156: t[b,c,h,w] = bit(h*S+w); rows n..batch zero-filled */
ORACLE_API int oracle_planes_to_tensor(const uint64_t *planes, uint32_t n, uint32_t C, uint32_t w64,
                                       uint32_t S, uint32_t batch, float *out) {
    if (n < 1 || n > batch || S < 1 || S > MAX_S || w64 * 64 < S * S) return -1;
    uint32_t hw = S * S;
    for (uint32_t b = 0; b < n; b++)
        for (uint32_t c = 0; c < C; c++) {
            const uint64_t *pl = planes + ((size_t)b * C + c) * w64;
            float *o = out + ((size_t)b * C + c) * hw;
            for (uint32_t i = 0; i < hw; i++) o[i] = plane_bit(pl, i) ? 1.0f : 0.0f;
        }
    for (size_t i = (size_t)n * C * hw; i < (size_t)batch * C * hw; i++) out[i] = 0.0f;
    return 0;
}

/* -------------------------------------------------------- BN folding --- */

static void fold_conv(const float *w, uint32_t cout, uint32_t cin, uint32_t taps, const float *gamma,
                      const float *beta, const float *mean, const float *var, float *wf, float *bf) {
    /* w: [cout][cin][taps] (torch OIHW) -> wf: [tap][cout][cin] */
    for (uint32_t co = 0; co < cout; co++) {
        float g = gamma ? gamma[co] : 1.0f;
        float be = beta ? beta[co] : 0.0f;
        float scale = g / sqrtf(var[co] + BN_EPS);
        bf[co] = be - mean[co] * scale;
        for (uint32_t ci = 0; ci < cin; ci++)
            for (uint32_t t = 0; t < taps; t++)
                wf[((size_t)t * cout + co) * cin + ci] = w[((size_t)co * cin + ci) * taps + t] * scale;
    }
}

static const float *take(const float **p, size_t n) {
    const float *r = *p;
    *p += n;
    return r;
}

static int make_conv3(conv3_layer *L, const float **p, uint32_t cout, uint32_t cin, int affine) {
    const float *w = take(p, (size_t)cout * cin * 9);
    const float *gamma = affine ? take(p, cout) : NULL;
    const float *beta = affine ? take(p, cout) : NULL;
    const float *mean = take(p, cout);
    const float *var = take(p, cout);
    L->cin = cin;
    L->cout = cout;
    L->w = (float *)malloc(sizeof(float) * 9 * cout * cin);
    L->b = (float *)malloc(sizeof(float) * cout);
    if (!L->w || !L->b) return -1;
    fold_conv(w, cout, cin, 9, gamma, beta, mean, var, L->w, L->b);
    return 0;
}

ORACLE_API size_t oracle_blob_nbytes(uint32_t C, uint32_t S, uint32_t M, uint32_t blocks, uint32_t F,
                                     uint32_t vhc, uint32_t phc) {
    size_t hw = (size_t)S * S, n = 0;
    if (F == 0) { /* SimpleTwoHeadedModel: two K x K dense layers, a 1 x K and an M x K head, K = C * hw */
        const size_t K = (size_t)C * hw;
        return HEADER_BYTES + 4 * (2 * (K * K + K) + K + 1 + (size_t)M * K + M);
    }
    n += (size_t)F * C * 9 + 4 * (size_t)F;
    n += (size_t)blocks * (2 * (size_t)F * F * 9 + 6 * (size_t)F);
    n += (size_t)vhc * F + 2 * (size_t)vhc + 128 * vhc * hw + 128 + 128 + 1;
    n += (size_t)phc * F + 2 * (size_t)phc + (size_t)M * phc * hw + M;
    return HEADER_BYTES + 4 * n;
}

ORACLE_API void oracle_net_destroy(oracle_net *net) {
    if (!net) return;
    free(net->stem.w);
    free(net->stem.b);
    for (uint32_t i = 0; net->c1 && i < net->d.blocks; i++) {
        free(net->c1[i].w);
        free(net->c1[i].b);
        free(net->c2[i].w);
        free(net->c2[i].b);
    }
    free(net->c1);
    free(net->c2);
    free(net->vconv_w);
    free(net->vconv_b);
    free(net->pconv_w);
    free(net->pconv_b);
    free(net->blob);
    free(net);
}

ORACLE_API oracle_net *oracle_net_create(const void *blob, size_t nbytes) {
    if (nbytes < HEADER_BYTES || memcmp(blob, "CATTUSW1", 8) != 0) return NULL;
    uint32_t h[9];
    memcpy(h, (const char *)blob + 8, sizeof h);
    if (h[0] != 1 || h[8] != 128) return NULL;
    oracle_desc d = {h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]};
    if (d.S < 1 || d.S > MAX_S || d.C < 1 || d.M < 1) return NULL;
    if (d.F != 0 && (d.vhc < 1 || d.phc < 1)) return NULL;
    if (nbytes != oracle_blob_nbytes(d.C, d.S, d.M, d.blocks, d.F, d.vhc, d.phc)) return NULL;

    oracle_net *net = (oracle_net *)calloc(1, sizeof *net);
    if (!net) return NULL;
    net->d = d;
    net->blob = (float *)malloc(nbytes - HEADER_BYTES);
    if (!net->blob) goto fail;
    memcpy(net->blob, (const char *)blob + HEADER_BYTES, nbytes - HEADER_BYTES);
    const float *p = net->blob;
    uint32_t hw = d.S * d.S;

    if (d.F == 0) { /* SimpleTwoHeadedModel: state_dict order _dense1, _dense2, _value_head, _policy_head */
        const size_t K = (size_t)d.C * hw;
        net->d1_w = take(&p, K * K), net->d1_b = take(&p, K);
        net->d2_w = take(&p, K * K), net->d2_b = take(&p, K);
        net->sv_w = take(&p, K), net->sv_b = take(&p, 1);
        net->sp_w = take(&p, (size_t)d.M * K), net->sp_b = take(&p, d.M);
        return net;
    }
    if (make_conv3(&net->stem, &p, d.F, d.C, 1)) goto fail;
    net->c1 = (conv3_layer *)calloc(d.blocks ? d.blocks : 1, sizeof(conv3_layer));
    net->c2 = (conv3_layer *)calloc(d.blocks ? d.blocks : 1, sizeof(conv3_layer));
    if (!net->c1 || !net->c2) goto fail;
    for (uint32_t i = 0; i < d.blocks; i++) {
        if (make_conv3(&net->c1[i], &p, d.F, d.F, 0)) goto fail;
        if (make_conv3(&net->c2[i], &p, d.F, d.F, 1)) goto fail;
    }
    {
        const float *w = take(&p, (size_t)d.vhc * d.F);
        const float *mean = take(&p, d.vhc), *var = take(&p, d.vhc);
        net->vconv_w = (float *)malloc(sizeof(float) * d.vhc * d.F);
        net->vconv_b = (float *)malloc(sizeof(float) * d.vhc);
        if (!net->vconv_w || !net->vconv_b) goto fail;
        fold_conv(w, d.vhc, d.F, 1, NULL, NULL, mean, var, net->vconv_w, net->vconv_b);
        net->vfc1_w = take(&p, (size_t)128 * d.vhc * hw);
        net->vfc1_b = take(&p, 128);
        net->vfc2_w = take(&p, 128);
        net->vfc2_b = take(&p, 1);
    }
    {
        const float *w = take(&p, (size_t)d.phc * d.F);
        const float *mean = take(&p, d.phc), *var = take(&p, d.phc);
        net->pconv_w = (float *)malloc(sizeof(float) * d.phc * d.F);
        net->pconv_b = (float *)malloc(sizeof(float) * d.phc);
        if (!net->pconv_w || !net->pconv_b) goto fail;
        fold_conv(w, d.phc, d.F, 1, NULL, NULL, mean, var, net->pconv_w, net->pconv_b);
        net->pfc_w = take(&p, (size_t)d.M * d.phc * hw);
        net->pfc_b = take(&p, d.M);
    }
    return net;
fail:
    oracle_net_destroy(net);
    return NULL;
}

/* ------------------------------------------------------------ layers --- */

static const int KPERM[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* One 3x3 'same' conv for one position in the canonical order.
 * in : [cin][S*S], out: [cout][S*S]; res (optional): [cout][S*S].
 * The board is held zero-padded with row pitch P = S+2; outputs are accumulated over the padded
 * flat index i = h*P + w (columns w >= S are scratch), so each (cout, k, tap) update is one
 * contiguous fmaf sweep (SWEEP below: 16-, 8- or 1-wide).  Every real output still sees exactly
 * the canonical chain: one fmaf per (chunk, tap, k) in that order, zero taps included; a vector
 * fma lane rounds exactly like fmaf. */
#define ACC_MAX (MAX_S * (MAX_S + 2) + 16)
/* Zeroed per-thread scratch for the padded input planes.  It is kept between calls: a calloc/free of
 * ~150 KB per layer goes through mmap/munmap, whose page faults serialise the threads of one process
 * (16 OpenMP threads then run no faster than one). */
static float *conv_scratch(size_t n) {
    static __thread float *buf = NULL;
    static __thread size_t cap = 0;
    if (n > cap) {
        free(buf);
        buf = (float *)malloc(n * sizeof(float));
        cap = buf ? n : 0;
    }
    if (buf) memset(buf, 0, n * sizeof(float));
    return buf;
}
#define CONV3X3_BODY(SWEEP)                                                                              \
    const uint32_t hw = S * S, P = S + 2, cin = L->cin, cout = L->cout;                                   \
    const uint32_t span = (S * P + 15) & ~15u;                                                            \
    const uint32_t plane = P * P + P + 32; /* slack: the last tap's sweep stays in bounds */              \
    float *xpad = conv_scratch((size_t)cin * plane);                                                      \
    for (uint32_t c = 0; c < cin; c++)                                                                    \
        for (uint32_t h = 0; h < S; h++)                                                                  \
            for (uint32_t w = 0; w < S; w++)                                                              \
                xpad[(size_t)c * plane + (h + 1) * P + (w + 1)] = in[c * hw + h * S + w];                 \
    const uint32_t nchunks = (cin + 31) / 32;                                                             \
    for (uint32_t co = 0; co < cout; co++) {                                                              \
        float acc[ACC_MAX] __attribute__((aligned(64)));                                                  \
        for (uint32_t i = 0; i < span; i++) acc[i] = 0.0f;                                                \
        for (uint32_t ch = 0; ch < nchunks; ch++)                                                         \
            for (uint32_t tap = 0; tap < 9; tap++) {                                                      \
                const uint32_t ty = tap / 3, tx = tap % 3; /* dy = ty-1, dx = tx-1 */                     \
                const float *wrow = L->w + ((size_t)tap * cout + co) * cin;                               \
                for (uint32_t kk = 0; kk < 32; kk++) {                                                    \
                    const uint32_t k = ch * 32 + (kk & ~7u) + (uint32_t)KPERM[kk & 7];                    \
                    if (k >= cin) continue;                                                               \
                    const float wv = wrow[k];                                                             \
                    const float *xp = xpad + (size_t)k * plane + ty * P + tx;                             \
                    SWEEP(acc, xp, wv, span);                                                             \
                }                                                                                         \
            }                                                                                             \
        const float b = L->b[co];                                                                         \
        for (uint32_t h = 0; h < S; h++)                                                                  \
            for (uint32_t w = 0; w < S; w++) {                                                            \
                const uint32_t i = h * S + w;                                                             \
                float y = acc[h * P + w] + b;                                                             \
                if (res) y = y + res[co * hw + i];                                                        \
                out[co * hw + i] = y > 0.0f ? y : 0.0f;                                                   \
            }                                                                                             \
    }

#define SWEEP_SCALAR(acc, xp, wv, span) \
    for (uint32_t i = 0; i < (span); i++) (acc)[i] = __builtin_fmaf((wv), (xp)[i], (acc)[i])
#define SWEEP_AVX2(acc, xp, wv, span)                                                                    \
    do {                                                                                                  \
        const __m256 wv8 = _mm256_set1_ps(wv);                                                            \
        for (uint32_t i = 0; i < (span); i += 8)                                                          \
            _mm256_store_ps((acc) + i, _mm256_fmadd_ps(wv8, _mm256_loadu_ps((xp) + i), _mm256_load_ps((acc) + i))); \
    } while (0)
#define SWEEP_AVX512(acc, xp, wv, span)                                                                  \
    do {                                                                                                  \
        const __m512 wv16 = _mm512_set1_ps(wv);                                                           \
        for (uint32_t i = 0; i < (span); i += 16)                                                         \
            _mm512_store_ps((acc) + i, _mm512_fmadd_ps(wv16, _mm512_loadu_ps((xp) + i), _mm512_load_ps((acc) + i))); \
    } while (0)

static void conv3x3_scalar(const conv3_layer *L, const float *in, const float *res, float *out, uint32_t S) {
    CONV3X3_BODY(SWEEP_SCALAR)
}
__attribute__((target("avx2,fma"))) static void conv3x3_avx2(const conv3_layer *L, const float *in,
                                                             const float *res, float *out, uint32_t S) {
    CONV3X3_BODY(SWEEP_AVX2)
}
__attribute__((target("avx512f"))) static void conv3x3_avx512(const conv3_layer *L, const float *in,
                                                              const float *res, float *out, uint32_t S) {
    CONV3X3_BODY(SWEEP_AVX512)
}

static void conv3x3(const conv3_layer *L, const float *in, const float *res, float *out, uint32_t S) {
    static int isa = -1; /* 0 scalar, 1 avx2+fma, 2 avx512f; all three give identical bits */
    if (isa < 0) {
        const char *force = getenv("ORACLE_ISA");
        __builtin_cpu_init();
        int v = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
        if (force) v = atoi(force) < v ? atoi(force) : v;
        isa = v;
    }
    if (isa == 2) conv3x3_avx512(L, in, res, out, S);
    else if (isa == 1) conv3x3_avx2(L, in, res, out, S);
    else conv3x3_scalar(L, in, res, out, S);
}

/* k index of the kk-th term of a dot product: 8-groups ascending, 0,4,1,5,2,6,3,7 inside a group */
static inline uint32_t kperm(uint32_t kk) { return (kk & ~7u) + (uint32_t)KPERM[kk & 7]; }

/* 1x1 conv + folded BN + ReLU: out[oc][p] = relu(chain_k(w[oc][k]*in[k][p]) + b[oc]) */
HOT static void conv1x1_relu(const float *w, const float *b, uint32_t oc_n, uint32_t F, uint32_t hw,
                             const float *in, float *out) {
    const uint32_t F8 = (F + 7) & ~7u;
    for (uint32_t oc = 0; oc < oc_n; oc++)
        for (uint32_t p = 0; p < hw; p++) {
            float acc = 0.0f;
            for (uint32_t kk = 0; kk < F8; kk++) {
                const uint32_t k = kperm(kk);
                if (k < F) acc = __builtin_fmaf(w[oc * F + k], in[k * hw + p], acc);
            }
            float y = acc + b[oc];
            out[oc * hw + p] = y > 0.0f ? y : 0.0f;
        }
}

/* y[j] = chain_k(w[j][k]*x[k]) + b[j] */
HOT static void linear(const float *w, const float *b, uint32_t out_n, uint32_t in_n, const float *x,
                       float *y) {
    const uint32_t n8 = (in_n + 7) & ~7u;
    for (uint32_t j = 0; j < out_n; j++) {
        float acc = 0.0f;
        const float *wr = w + (size_t)j * in_n;
        for (uint32_t kk = 0; kk < n8; kk++) {
            const uint32_t k = kperm(kk);
            if (k < in_n) acc = __builtin_fmaf(wr[k], x[k], acc);
        }
        y[j] = acc + b[j];
    }
}

/* Forward one position. tower_out (optional): [F][hw] activations after the residual tower;
 * stem_out (optional): [F][hw] after the stem. */
static int forward_one(const oracle_net *net, const uint64_t *planes, uint32_t w64, float *policy,
                       float *value, float *stem_out, float *tower_out) {
    const oracle_desc *d = &net->d;
    const uint32_t hw = d->S * d->S, F = d->F;
    if (F == 0) {
        /* SimpleTwoHeadedModel.forward (net_utils.py:112-121): flatten (C-major: c * hw + p) -> dense + ReLU -> dense + ReLU ->
         * value: dense + tanh; policy: dense (raw logits) + the scrub of net/mod.rs:56-61.  Same chain order as every other
         * dot product of this file (8-groups of k ascending, 0,4,1,5,2,6,3,7 inside). */
        if (stem_out || tower_out) return -1;
        const uint32_t K = d->C * hw;
        float *x = (float *)calloc(3 * (size_t)K, sizeof(float));
        if (!x) return -1;
        float *h1 = x + K, *h2 = x + 2 * K, v;
        for (uint32_t c = 0; c < d->C; c++)
            for (uint32_t i = 0; i < hw; i++) x[c * hw + i] = plane_bit(planes + (size_t)c * w64, i) ? 1.0f : 0.0f;
        linear(net->d1_w, net->d1_b, K, K, x, h1);
        for (uint32_t j = 0; j < K; j++) h1[j] = h1[j] > 0.0f ? h1[j] : 0.0f;
        linear(net->d2_w, net->d2_b, K, K, h1, h2);
        for (uint32_t j = 0; j < K; j++) h2[j] = h2[j] > 0.0f ? h2[j] : 0.0f;
        linear(net->sv_w, net->sv_b, 1, K, h2, &v);
        *value = oracle_tanhf(v);
        linear(net->sp_w, net->sp_b, d->M, K, h2, policy);
        for (uint32_t m = 0; m < d->M; m++)
            if (!isfinite(policy[m])) policy[m] = -3.40282347e+38f; /* f32::MIN */
        free(x);
        return 0;
    }
    float *x0 = (float *)malloc(sizeof(float) * d->C * hw);
    float *a = (float *)malloc(sizeof(float) * F * hw);
    float *t = (float *)malloc(sizeof(float) * F * hw);
    float *y = (float *)malloc(sizeof(float) * F * hw);
    float *hv = (float *)malloc(sizeof(float) * (d->vhc + d->phc) * hw);
    if (!x0 || !a || !t || !y || !hv) return -1;
    for (uint32_t c = 0; c < d->C; c++)
        for (uint32_t i = 0; i < hw; i++) x0[c * hw + i] = plane_bit(planes + (size_t)c * w64, i) ? 1.0f : 0.0f;

    conv3x3(&net->stem, x0, NULL, a, d->S);
    if (stem_out) memcpy(stem_out, a, sizeof(float) * F * hw);
    for (uint32_t i = 0; i < d->blocks; i++) {
        conv3x3(&net->c1[i], a, NULL, t, d->S);
        conv3x3(&net->c2[i], t, a, y, d->S);
        float *s = a;
        a = y;
        y = s;
    }
    if (tower_out) memcpy(tower_out, a, sizeof(float) * F * hw);

    /* value head: net_utils.py:68-75 (flatten is C-major: c*hw + p) */
    float h1[128], v;
    conv1x1_relu(net->vconv_w, net->vconv_b, d->vhc, F, hw, a, hv);
    linear(net->vfc1_w, net->vfc1_b, 128, d->vhc * hw, hv, h1);
    for (int j = 0; j < 128; j++) h1[j] = h1[j] > 0.0f ? h1[j] : 0.0f;
    linear(net->vfc2_w, net->vfc2_b, 1, 128, h1, &v);
    *value = oracle_tanhf(v);

    /* policy head: net_utils.py:78-82 (raw logits) + scrub net/mod.rs:56-61 */
    float *hp = hv + d->vhc * hw;
    conv1x1_relu(net->pconv_w, net->pconv_b, d->phc, F, hw, a, hp);
    linear(net->pfc_w, net->pfc_b, d->M, d->phc * hw, hp, policy);
    for (uint32_t m = 0; m < d->M; m++)
        if (!isfinite(policy[m])) policy[m] = -3.40282347e+38f; /* f32::MIN */

    free(x0);
    free(a);
    free(t);
    free(y);
    free(hv);
    return 0;
}

/* planes: [n][C][w64] u64 (bit i of a plane = bit i&63 of word i>>6); policy: [n][M]; value: [n].
 * threads<=0 -> all cores. */
ORACLE_API int oracle_net_forward(const oracle_net *net, const uint64_t *planes, uint32_t w64, uint32_t n,
                                  float *policy, float *value, int threads) {
    if (!net || !planes || !policy || !value) return -1;
    const oracle_desc *d = &net->d;
    if (w64 * 64 < d->S * d->S) return -1;
    const size_t stride = (size_t)d->C * w64;
    int rc = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int64_t b = 0; b < (int64_t)n; b++) {
        if (forward_one(net, planes + b * stride, w64, policy + b * d->M, value + b, NULL, NULL)) rc = -1;
    }
    (void)threads;
    return rc;
}

/* The same behind the network-callback signature of the host library (cattus_net_eval_fn,
 * include/cattus_selfplay.h), so that the C++ search can run on the CPU oracle with no Python in the
 * loop: bench.py's cpu_baseline leg and the parity tests.  ctx -> {net, plane words, threads}. */
typedef struct oracle_cb_ctx {
    const oracle_net *net;
    uint32_t w64;
    int32_t threads;
} oracle_cb_ctx;
ORACLE_API int oracle_net_eval_cb(void *ctx, const uint64_t *planes, uint32_t n, float *policy, float *value) {
    const oracle_cb_ctx *c = (const oracle_cb_ctx *)ctx;
    if (!c) return -1;
    return oracle_net_forward(c->net, planes, c->w64, n, policy, value, c->threads);
}

/* Single position with intermediate activations, NCHW [F][hw] each. */
ORACLE_API int oracle_net_forward_debug(const oracle_net *net, const uint64_t *planes, uint32_t w64,
                                        float *policy, float *value, float *stem_out, float *tower_out) {
    if (!net || w64 * 64 < net->d.S * net->d.S) return -1;
    return forward_one(net, planes, w64, policy, value, stem_out, tower_out);
}

ORACLE_API void oracle_net_desc(const oracle_net *net, uint32_t out[8]) {
    memcpy(out, &net->d, sizeof(oracle_desc));
}

/* Folded tensors, for cross-checking the evaluator's own folding.
 * which: 0 = stem, 1+2i = block i conv1, 2+2i = block i conv2. */
ORACLE_API int oracle_net_folded_conv(const oracle_net *net, uint32_t which, float *w_out, float *b_out) {
    const conv3_layer *L;
    if (which == 0) L = &net->stem;
    else {
        uint32_t i = (which - 1) / 2;
        if (i >= net->d.blocks) return -1;
        L = ((which - 1) % 2 == 0) ? &net->c1[i] : &net->c2[i];
    }
    memcpy(w_out, L->w, sizeof(float) * 9 * L->cout * L->cin);
    memcpy(b_out, L->b, sizeof(float) * L->cout);
    return 0;
}

/* engine/src/net/mod.rs:106-119: gather logits at idx, max-subtract from f32::MIN fold,
 * exp, sequential f32 sum, divide. */
ORACLE_API void oracle_softmax_legal(const float *logits, const uint32_t *idx, uint32_t k, float *probs) {
    float max_p = -3.40282347e+38f;
    for (uint32_t i = 0; i < k; i++) {
        float s = logits[idx[i]];
        max_p = s > max_p ? s : max_p; /* f32::max semantics for non-NaN input */
    }
    float sum = 0.0f;
    for (uint32_t i = 0; i < k; i++) {
        probs[i] = expf(logits[idx[i]] - max_p);
        sum += probs[i];
    }
    for (uint32_t i = 0; i < k; i++) probs[i] = probs[i] / sum;
}

/* The same softmax with the evaluator's own exp (arguments below -80 give 0): the restatement that
 * cattus_hip_eval_legal's device kernel is held to bit for bit.  Differs from oracle_softmax_legal
 * only by libm's expf rounding (a few ulp). */
ORACLE_API void oracle_softmax_legal_det(const float *logits, const uint16_t *idx, uint32_t k, float *probs) {
    float max_p = -3.40282347e+38f;
    for (uint32_t i = 0; i < k; i++) {
        float s = logits[idx[i]];
        max_p = s > max_p ? s : max_p;
    }
    float sum = 0.0f;
    for (uint32_t i = 0; i < k; i++) {
        float x = logits[idx[i]] - max_p;
        probs[i] = x < -80.0f ? 0.0f : oracle_expf_pos(x);
        sum += probs[i];
    }
    for (uint32_t i = 0; i < k; i++) probs[i] = probs[i] / sum;
}
