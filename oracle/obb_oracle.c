/*
 * TEST INFRASTRUCTURE -- CPU oracle for the oriented-box hot path.
 *
 * This library is the *checker* for the HIP kernels in yolov5_obb_amd/csrc.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product path never does (see oracle/README.md).
 *
 * It restates, in plain C, the algorithms of the reference repository
 * hukaixuan19970627/yolov5_obb (every function cites the file:line it
 * follows).  Parity status: PINNED -- validated bit-for-bit against the
 * reference's own sources compiled in the build container (oracle/_ref, see
 * oracle/Makefile and tests/test_oracle_vs_ref.py) and against the frozen
 * fixtures in tests/golden/ produced from them.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction; the
 * reference's CPU build on x86-64 has none either).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- rotated-rect IoU, float and double flavours -------------------------- */
#define REAL float
#define SFX(n) n##_f32
#include "riou_impl.inc"
#undef REAL
#undef SFX
#define REAL double
#define SFX(n) n##_f64
#include "riou_impl.inc"
#undef REAL
#undef SFX

/* ---- quad IoU: float = CUDA device flavour, double = polyiou.cpp ----------- */
#define REAL float
#define SFX(n) n##_f32
#define PIOU_DEGENERATE_RULE 1
#include "piou_impl.inc"
#undef REAL
#undef SFX
#undef PIOU_DEGENERATE_RULE
#define REAL double
#define SFX(n) n##_f64
#define PIOU_DEGENERATE_RULE 0
#include "piou_impl.inc"
#undef REAL
#undef SFX
#undef PIOU_DEGENERATE_RULE

/* ---- ordering --------------------------------------------------------------
 * scores.sort(0, descending=True)  (nms_rotated_cuda.cu:81, nms_rotated_cpu.cpp:27,
 * poly_nms_cuda.cu:204).  torch's sort is stable in practice on both devices;
 * the documented tie rule of this project is: equal scores keep ascending
 * original index, NaN sorts first (torch treats NaN as the largest value). */
static int score_before_f32(float a, float b)   /* strictly before in descending order */
{
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an && !bn;
    return a > b;
}
static int score_before_f64(double a, double b)
{
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an && !bn;
    return a > b;
}

#define DEFINE_ORDER(NAME, T, BEFORE)                                              \
    void NAME(const T *scores, int64_t n, int64_t *order)                          \
    {                                                                              \
        int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));   \
        for (int64_t i = 0; i < n; i++) order[i] = i;                              \
        for (int64_t w = 1; w < n; w *= 2) {            /* bottom-up merge sort */ \
            for (int64_t lo = 0; lo < n; lo += 2 * w) {                            \
                int64_t mid = lo + w < n ? lo + w : n;                             \
                int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;                      \
                int64_t a = lo, b = mid, k = lo;                                   \
                while (a < mid && b < hi)                                          \
                    tmp[k++] = BEFORE(scores[order[b]], scores[order[a]])          \
                                   ? order[b++] : order[a++];                      \
                while (a < mid) tmp[k++] = order[a++];                             \
                while (b < hi) tmp[k++] = order[b++];                              \
            }                                                                      \
            memcpy(order, tmp, sizeof(int64_t) * (size_t)n);                       \
        }                                                                          \
        free(tmp);                                                                 \
    }
DEFINE_ORDER(oracle_order_desc_f32, float, score_before_f32)
DEFINE_ORDER(oracle_order_desc_f64, double, score_before_f64)

/* ---- greedy NMS --------------------------------------------------------------
 * Equivalent to the mask + host scan of nms_rotated_cuda.cu:109-128 (a box is
 * dropped iff an earlier *kept* box has IoU(kept, box) > thr; the IoU argument
 * order is (higher-scored row box, lower-scored column box), :60) and to the
 * double loop of nms_rotated_cpu.cpp:36-58.
 *   ge == 0 : strict  ">"   -- the CUDA path   (nms_rotated_cuda.cu:60)
 *   ge == 1 : ">="          -- the CPU path    (nms_rotated_cpu.cpp:55)
 * Returns the number kept; keep[] holds original indices in score order. */
/* oracle_set_threads(t): t > 1 splits the INNER loop (one kept box against the later boxes) over t OpenMP threads.
 * Every iteration of that loop only reads box a and writes dead[b] of its own b, so the result is the same for any
 * thread count; the option exists so that the 100k S-uniform case (1.8e9 IoU calls) finishes in a test's time. */
static int g_threads = 1;
void oracle_set_threads(int t) { g_threads = t > 1 ? t : 1; }
int oracle_get_threads(void) { return g_threads; }

#define DEFINE_NMS(NAME, T, ORDER, IOU, STRIDE)                                     \
    int64_t NAME(const T *dets, const T *scores, int64_t n, T thr, int ge,          \
                 int64_t *keep)                                                     \
    {                                                                               \
        if (n <= 0) return 0;                                                       \
        int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);            \
        uint8_t *dead = (uint8_t *)calloc((size_t)n, 1);                            \
        ORDER(scores, n, order);                                                    \
        int64_t nk = 0;                                                             \
        const int nth = g_threads;                                                  \
        for (int64_t a = 0; a < n; a++) {                                           \
            if (dead[a]) continue;                                                  \
            int64_t i = order[a];                                                   \
            keep[nk++] = i;                                                         \
            _Pragma("omp parallel for schedule(static) num_threads(nth) if (nth > 1 && n - a > 2048)") \
            for (int64_t b = a + 1; b < n; b++) {                                   \
                if (dead[b]) continue;                                              \
                T v = IOU(dets + i * STRIDE, dets + order[b] * STRIDE);             \
                if (ge ? (v >= thr) : (v > thr)) dead[b] = 1;                       \
            }                                                                       \
        }                                                                           \
        free(order);                                                                \
        free(dead);                                                                 \
        return nk;                                                                  \
    }
DEFINE_NMS(oracle_nms_rotated_f32, float, oracle_order_desc_f32, oracle_riou_f32, 5)
DEFINE_NMS(oracle_nms_rotated_f64, double, oracle_order_desc_f64, oracle_riou_f64, 5)

/* poly NMS: rows are 8 coordinates + score (poly_nms_cuda.cu:197-261, strict ">" :187) */
int64_t oracle_nms_poly_f32(const float *polys9, int64_t n, float thr, int64_t *keep)
{
    if (n <= 0) return 0;
    float *sc = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    uint8_t *dead = (uint8_t *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; i++) sc[i] = polys9[i * 9 + 8];
    oracle_order_desc_f32(sc, n, order);
    int64_t nk = 0;
    for (int64_t a = 0; a < n; a++) {
        if (dead[a]) continue;
        int64_t i = order[a];
        keep[nk++] = i;
        for (int64_t b = a + 1; b < n; b++) {
            if (dead[b]) continue;
            if (oracle_piou_f32(polys9 + i * 9, polys9 + order[b] * 9) > thr) dead[b] = 1;
        }
    }
    free(sc); free(order); free(dead);
    return nk;
}

/* devkit flavour: the caller pre-sorts (poly_nms.pyx:18-21), the C function
 * scans rows in the given order and returns *positions*
 * (poly_nms_kernel.cu:277-329). */
int oracle_devkit_poly_nms(int *keep_out, const float *polys, int n, int dim, float thr)
{
    uint8_t *dead = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int a = 0; a < n; a++) {
        if (dead[a]) continue;
        keep_out[nk++] = a;
        for (int b = a + 1; b < n; b++) {
            if (dead[b]) continue;
            if (oracle_piou_f32(polys + (size_t)a * dim, polys + (size_t)b * dim) > thr) dead[b] = 1;
        }
    }
    free(dead);
    return nk;
}

/* ---- dense pairwise matrices ------------------------------------------------ */
void oracle_riou_matrix_f32(const float *a5, int64_t n, const float *b5, int64_t k, float *out)
{
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < k; j++) out[i * k + j] = oracle_riou_f32(a5 + i * 5, b5 + j * 5);
}

void oracle_piou_matrix_f32(const float *a8, int64_t n, int64_t sa, const float *b8, int64_t k, int64_t sb,
                            float *out)
{
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < k; j++) out[i * k + j] = oracle_piou_f32(a8 + i * sa, b8 + j * sb);
}

void oracle_piou_matrix_f64(const double *a8, int64_t n, const double *b8, int64_t k, double *out)
{
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < k; j++) out[i * k + j] = oracle_piou_f64(a8 + i * 8, b8 + j * 8);
}

/* rbox [cx,cy,w,h,theta] -> quad, as DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-297:
 * float cos/sin, corner arithmetic in double (the "/ 2.0" literals), one rounding to float.
 * The reference's cos(float) is CUDA's device cosf (<= 2 ulp, not correctly rounded, not reproducible elsewhere); the
 * restatement takes the correctly rounded float (double cos / sin rounded once), as the product's kernel does. */
void oracle_rbox2quad_devkit(const float *b, float *q8)
{
    float cs = (float)cos((double)b[4]), ss = (float)sin((double)b[4]);
    double w = b[2], h = b[3], x = b[0], y = b[1];
    q8[0] = (float)(x + cs * (w / 2.0) - ss * (-h / 2.0));
    q8[2] = (float)(x + cs * (w / 2.0) - ss * (h / 2.0));
    q8[4] = (float)(x + cs * (-w / 2.0) - ss * (h / 2.0));
    q8[6] = (float)(x + cs * (-w / 2.0) - ss * (-h / 2.0));
    q8[1] = (float)(y + ss * (w / 2.0) + cs * (-h / 2.0));
    q8[3] = (float)(y + ss * (w / 2.0) + cs * (h / 2.0));
    q8[5] = (float)(y + ss * (-w / 2.0) + cs * (h / 2.0));
    q8[7] = (float)(y + ss * (-w / 2.0) + cs * (-h / 2.0));
}

/* overlaps_kernel (poly_overlaps_kernel.cu:330-353): out[n*K + k] */
void oracle_devkit_overlaps(float *out, const float *boxes, const float *query, int n, int k)
{
    float *qb = (float *)malloc(sizeof(float) * 8 * (size_t)(k > 0 ? k : 1));
    for (int j = 0; j < k; j++) oracle_rbox2quad_devkit(query + 5 * j, qb + 8 * j);
    for (int i = 0; i < n; i++) {
        float pa[8];
        oracle_rbox2quad_devkit(boxes + 5 * i, pa);
        for (int j = 0; j < k; j++) out[(size_t)i * k + j] = oracle_piou_f32(pa, qb + 8 * j);
    }
    free(qb);
}
