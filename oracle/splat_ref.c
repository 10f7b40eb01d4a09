/* TEST INFRASTRUCTURE ONLY -- scalar C restatement of the reference's three forward-splat
 * kernels for {{type}} = float (SS = algorithms/diffusion_animation/softsplat_new.py):
 *   ofd_ref_splat_out       SS:352-423  (softsplat_out)
 *   ofd_ref_splat_ingrad    SS:489-565  (softsplat_ingrad)
 *   ofd_ref_splat_flowgrad  SS:600-700  (softsplat_flowgrad)
 * One loop iteration == one CUDA thread of the reference.  The arithmetic keeps the
 * reference's float/double mix: the bare `1.0` / `0.0` literals are double, so the remap
 * bracket is evaluated in double and rounded to float on assignment.
 * Tensors are contiguous NCHW.  PARITY UNPINNED: the reference kernels cannot execute
 * without CUDA+cupy (SS:444), see oracle/__init__.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int x0, y0;       /* north-west corner */
    float fx, fy;     /* remapped sample position */
    float dxx, dyy;   /* flowgrad "frozen" factors (SS:626-647) */
    int ok;
} sample_t;

/* variant: 0 = forward (SS:374-390), 1 = ingrad (SS:515-533), 2 = flowgrad (SS:628-647) */
static sample_t remap(float flow_x, float flow_y, int x, int y, int H, int W,
                      int scale, int ox, int oy, int variant) {
    sample_t s;
    memset(&s, 0, sizeof s);
    float fltX = (float)x + flow_x;
    float fltY = (float)y + flow_y;
    if (!isfinite(fltX) || !isfinite(fltY)) return s;
    s.ok = 1;

    int guard = (variant == 0) ? (scale > 1) : 1;
    if ((double)fltX >= (double)(float)W - 1.0 && guard) {
        fltX = (float)((double)fltX + ((double)(fltX - (float)W) + 1.0) * (double)(float)((abs(ox - (W % scale))) % scale));
        if (variant == 1)
            fltX = (float)((double)fltX + ((double)(fltX - (float)W) + 1.0) * (double)(float)ox);
        fltX = (fltX - (float)ox) / (float)scale;
    } else if ((double)(fltX - (float)ox) < 0.0) {
        fltX = fltX - (float)ox;
    } else {
        fltX = (fltX - (float)ox) / (float)scale;
        s.dxx = 1.0f / (float)scale;
    }

    if ((double)fltY >= (double)(float)H - 1.0 && guard) {
        float mult = (variant == 2) ? (float)oy : (float)((abs(oy - (H % scale))) % scale);
        fltY = (float)((double)fltY + ((double)(fltY - (float)H) + 1.0) * (double)mult);
        fltY = (fltY - (float)oy) / (float)scale;
    } else if ((double)(fltY - (float)oy) < 0.0) {
        fltY = fltY - (float)oy;
    } else {
        fltY = (fltY - (float)oy) / (float)scale;
        s.dyy = 1.0f / (float)scale;
    }
    s.fx = fltX;
    s.fy = fltY;
    s.x0 = (int)floorf(fltX);
    s.y0 = (int)floorf(fltY);
    return s;
}

static inline int inside(int cx, int cy, int Wo, int Ho) {
    return cx >= 0 && cx < Wo && cy >= 0 && cy < Ho;
}

/* out must be zero-filled by the caller (SS:343-345). Optional corner dump (int32 x0,y0 per
 * source pixel of channel 0; INT_MIN-like -2^30 when skipped) for index-exactness tests. */
void ofd_ref_splat_out(const float* in, const float* flow, float* out, int* corners,
                       int B, int C, int H, int W, int scale, int ox, int oy) {
    const int Ho = H / scale, Wo = W / scale;
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const long fi = ((long)n * 2 * H + y) * W + x;
                    sample_t s = remap(flow[fi], flow[fi + (long)H * W], x, y, H, W, scale, ox, oy, 0);
                    if (c == 0 && corners) {
                        long ci = (((long)n * H + y) * W + x) * 2;
                        corners[ci] = s.ok ? s.x0 : -(1 << 30);
                        corners[ci + 1] = s.ok ? s.y0 : -(1 << 30);
                    }
                    if (!s.ok) continue;
                    const float v = in[(((long)n * C + c) * H + y) * W + x];
                    const int x0 = s.x0, y0 = s.y0, x1 = x0 + 1, y1 = y0 + 1;
                    const float wnw = ((float)x1 - s.fx) * ((float)y1 - s.fy);
                    const float wne = (s.fx - (float)x0) * ((float)y1 - s.fy);
                    const float wsw = ((float)x1 - s.fx) * (s.fy - (float)y0);
                    const float wse = (s.fx - (float)x0) * (s.fy - (float)y0);
                    float* o = out + ((long)n * C + c) * Ho * Wo;
                    if (inside(x0, y0, Wo, Ho)) o[(long)y0 * Wo + x0] += v * wnw;
                    if (inside(x1, y0, Wo, Ho)) o[(long)y0 * Wo + x1] += v * wne;
                    if (inside(x0, y1, Wo, Ho)) o[(long)y1 * Wo + x0] += v * wsw;
                    if (inside(x1, y1, Wo, Ho)) o[(long)y1 * Wo + x1] += v * wse;
                }
}

/* ingrad must be zero-filled by the caller (skipped samples keep 0, SS:468-474). */
void ofd_ref_splat_ingrad(const float* flow, const float* outgrad, float* ingrad,
                          int B, int C, int H, int W, int scale, int ox, int oy) {
    const int Ho = H / scale, Wo = W / scale;
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const long fi = ((long)n * 2 * H + y) * W + x;
                    sample_t s = remap(flow[fi], flow[fi + (long)H * W], x, y, H, W, scale, ox, oy, 1);
                    if (!s.ok) continue;
                    const int x0 = s.x0, y0 = s.y0, x1 = x0 + 1, y1 = y0 + 1;
                    const float wnw = ((float)x1 - s.fx) * ((float)y1 - s.fy);
                    const float wne = (s.fx - (float)x0) * ((float)y1 - s.fy);
                    const float wsw = ((float)x1 - s.fx) * (s.fy - (float)y0);
                    const float wse = (s.fx - (float)x0) * (s.fy - (float)y0);
                    const float* g = outgrad + ((long)n * C + c) * Ho * Wo;
                    float acc = 0.0f;
                    if (inside(x0, y0, Wo, Ho)) acc += g[(long)y0 * Wo + x0] * wnw;
                    if (inside(x1, y0, Wo, Ho)) acc += g[(long)y0 * Wo + x1] * wne;
                    if (inside(x0, y1, Wo, Ho)) acc += g[(long)y1 * Wo + x0] * wsw;
                    if (inside(x1, y1, Wo, Ho)) acc += g[(long)y1 * Wo + x1] * wse;
                    ingrad[(((long)n * C + c) * H + y) * W + x] = acc;
                }
}

/* flowgrad must be zero-filled by the caller.  Reproduces the crossed factors: channel 0
 * (d/dflow_x) is scaled by dfltYY and channel 1 by dfltXX (SS:664-665, SS:671-672). */
void ofd_ref_splat_flowgrad(const float* in, const float* flow, const float* outgrad, float* flowgrad,
                            int B, int C, int H, int W, int scale, int ox, int oy) {
    const int Ho = H / scale, Wo = W / scale;
    for (int n = 0; n < B; ++n)
        for (int fc = 0; fc < 2; ++fc)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const long fi = ((long)n * 2 * H + y) * W + x;
                    sample_t s = remap(flow[fi], flow[fi + (long)H * W], x, y, H, W, scale, ox, oy, 2);
                    if (!s.ok) continue;
                    const int x0 = s.x0, y0 = s.y0, x1 = x0 + 1, y1 = y0 + 1;
                    float wnw, wne, wsw, wse, d;
                    if (fc == 0) {
                        d = s.dyy;
                        wnw = -1.0f * ((float)y1 - s.fy);
                        wne = +1.0f * ((float)y1 - s.fy);
                        wsw = -1.0f * (s.fy - (float)y0);
                        wse = +1.0f * (s.fy - (float)y0);
                    } else {
                        d = s.dxx;
                        wnw = ((float)x1 - s.fx) * -1.0f;
                        wne = (s.fx - (float)x0) * -1.0f;
                        wsw = ((float)x1 - s.fx) * +1.0f;
                        wse = (s.fx - (float)x0) * +1.0f;
                    }
                    float acc = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        const float v = in[(((long)n * C + c) * H + y) * W + x];
                        const float* g = outgrad + ((long)n * C + c) * Ho * Wo;
                        if (inside(x0, y0, Wo, Ho)) acc += g[(long)y0 * Wo + x0] * v * wnw * d;
                        if (inside(x1, y0, Wo, Ho)) acc += g[(long)y0 * Wo + x1] * v * wne * d;
                        if (inside(x0, y1, Wo, Ho)) acc += g[(long)y1 * Wo + x0] * v * wsw * d;
                        if (inside(x1, y1, Wo, Ho)) acc += g[(long)y1 * Wo + x1] * v * wse * d;
                    }
                    flowgrad[(((long)n * 2 + fc) * H + y) * W + x] = acc;
                }
}
