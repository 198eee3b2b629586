/* CPU restatement of the reference's per-image colour correction and scale-factor resize -- TEST INFRASTRUCTURE ONLY.
 *
 * (1) oracle_color_correct_u8 follows NeRSembleDataset.apply_color_correction, vhap/data/nersemble_dataset.py:165-171:
 *         rgb = item["rgb"] / 255                                         (uint8 / int -> float64 true divide)
 *         rgb = rgb @ A[:3, :3] + A[np.newaxis, :3, 3]                    (float64 matmul, then the translation column)
 *         item["rgb"] = (np.clip(rgb, 0, 1) * 255).astype(np.uint8)       (truncating cast)
 *     numpy hands the [H,W,3] @ [3,3] product to its BLAS (OpenBLAS dgemm on every FMA-capable x86-64): per output channel j the
 *     kernel accumulates  acc = r*A[0][j];  acc = fma(g, A[1][j], acc);  acc = fma(b, A[2][j], acc)  -- verified against numpy 2.2 /
 *     OpenBLAS 0.3.29 in the build container with exact rational arithmetic (tools/make_golden_ingest.py) and pinned by the golden
 *     table the reference's own method produced (tests/golden/ingest_cc_golden.npz).
 *
 * (2) oracle_pil_resize_u8 follows VideoDataset.apply_scale_factor, vhap/data/video_dataset.py:266-300:
 *         Image.fromarray(rgb).resize((w, h), resample=Image.BILINEAR)    -- and the same for the alpha map (:293-296)
 *     The arithmetic is Pillow's (third-party, requirement `Pillow`, unpinned in the reference; 12.2.0 in this image), whose published
 *     algorithm (src/libImaging/Resample.c) is restated here: precompute_coeffs() -- a triangle filter whose support grows with the
 *     downscale factor, coefficients normalised in double --, normalize_coeffs_8bpc() -- 22-bit fixed point, round half away from zero --,
 *     a horizontal pass into 8-bit, then a vertical pass, each `clip8((1 << 21) + sum) = clamp(sum >> 22, 0, 255)`.  Checked against
 *     PIL itself in tests/test_ingest.py wherever PIL is importable, and pinned by golden vectors made by the reference's method. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION_BITS (32 - 8 - 2)

/* A: the camera's affine colour transform, row-major with leading dimension ld (4 for the 4x4 npy files, 3-row 3x4 works too) */
void oracle_color_correct_u8(const uint8_t* rgb, long npx, const double* A, int ld, uint8_t* out) {
    for (long p = 0; p < npx; p++) {
        const double r = rgb[3 * p] / 255.0, g = rgb[3 * p + 1] / 255.0, b = rgb[3 * p + 2] / 255.0;
        for (int j = 0; j < 3; j++) {
            double acc = r * A[0 * ld + j];
            acc = fma(g, A[1 * ld + j], acc);
            acc = fma(b, A[2 * ld + j], acc);
            double v = acc + A[j * ld + 3];
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            out[3 * p + j] = (uint8_t)(v * 255.0);
        }
    }
}

static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

/* Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, inSize): returns ksize; bounds [2 * outSize], kk [outSize * ksize] */
int oracle_pil_coeffs(int inSize, int outSize, int* bounds, int32_t* kk, int kk_cap) {
    const float in0 = 0.0f, in1 = (float)inSize;
    double scale, filterscale, support;
    filterscale = scale = (double)(in1 - in0) / outSize;
    if (filterscale < 1.0) filterscale = 1.0;
    support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if ((long)outSize * ksize > kk_cap) return -ksize;
    double* k = (double*)malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < outSize; xx++) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            const double w = bilinear_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        for (x = 0; x < ksize; x++)
            kk[(long)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(k);
    return ksize;
}

static uint8_t clip8(int in) {
    const int v = in >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* src [H,W,C] -> dst [h,w,C] as Image.resize((w, h), BILINEAR) does for 8-bit images: horizontal pass (if w != W), then vertical (if h != H) */
int oracle_pil_resize_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int h, int w) {
    if (h == H && w == W) { memcpy(dst, src, (size_t)H * W * C); return 0; }
    const uint8_t* cur = src;
    uint8_t* tmp = NULL;
    int curW = W;
    if (w != W) {
        const double sc = (double)W / w;
        const int kmax = ((int)ceil(sc < 1.0 ? 1.0 : sc) * 2 + 1);
        int* bounds = (int*)malloc(sizeof(int) * 2 * w);
        int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * kmax);
        const int ks = oracle_pil_coeffs(W, w, bounds, kk, w * kmax);
        if (ks <= 0) return 1;
        tmp = (uint8_t*)malloc((size_t)H * w * C);
        for (int y = 0; y < H; y++)
            for (int xx = 0; xx < w; xx++) {
                const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
                for (int c = 0; c < C; c++) {
                    int ss = 1 << (PRECISION_BITS - 1);
                    for (int x = 0; x < xmax; x++) ss += src[((size_t)y * W + x + xmin) * C + c] * kk[(size_t)xx * ks + x];
                    tmp[((size_t)y * w + xx) * C + c] = clip8(ss);
                }
            }
        free(bounds); free(kk);
        cur = tmp;
        curW = w;
    }
    if (h != H) {
        const double sc = (double)H / h;
        const int kmax = ((int)ceil(sc < 1.0 ? 1.0 : sc) * 2 + 1);
        int* bounds = (int*)malloc(sizeof(int) * 2 * h);
        int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)h * kmax);
        const int ks = oracle_pil_coeffs(H, h, bounds, kk, h * kmax);
        if (ks <= 0) return 1;
        for (int yy = 0; yy < h; yy++) {
            const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
            for (int x = 0; x < curW * C; x++) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int y = 0; y < ymax; y++) ss += cur[(size_t)(y + ymin) * curW * C + x] * kk[(size_t)yy * ks + y];
                dst[(size_t)yy * curW * C + x] = clip8(ss);
            }
        }
        free(bounds); free(kk);
    } else {
        memcpy(dst, cur, (size_t)H * curW * C);
    }
    free(tmp);
    return 0;
}
