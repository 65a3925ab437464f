// post.hip -- the image post-process chain of the reference (PostProcessStack.cs and
// the shaders under Assets/Shaders/PostProcess) and SaveTexture's 8-bit conversion, as HIP
// image kernels over whole float4 frames in device memory.
//
// Every stage is one pass per destination pixel, memory-bound and tiny next to
// the path tracer (a 1080p bloom moves ~0.4 GB in 15 launches).  The kernels keep
// the reference's arithmetic order so the CPU checker can be matched bit for bit
// (include/rt_post.h fixes what the reference leaves to the sampler hardware).
#include <math.h>

#include <vector>

#include "rt_internal.h"
#include "../../include/rt_post.h"

struct RtImage {
    float4 *d = nullptr;
    int w = 0, h = 0;
    size_t cap = 0;   // pixels allocated
};

struct RtPostState {
    RtImage post, user, staged, lut, thr;
    std::vector<RtImage> downs, ups;
    float *dWeights = nullptr;      // 2 x 15 x 15 tap weights (down, up)
    uint32_t *dSrgbTable = nullptr; // 256 entries
    uint8_t *dRgb = nullptr;
    size_t rgbCap = 0;
};

static int image_reserve(RtContext *ctx, RtImage *im, int w, int h)
{
    const size_t px = (size_t)w * (size_t)h;
    if (px > im->cap) {
        if (im->d) (void)hipFree(im->d);
        im->d = nullptr;
        im->cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&im->d, px * sizeof(float4)));
        im->cap = px;
    }
    im->w = w;
    im->h = h;
    return RT_OK;
}

static void image_free(RtImage *im)
{
    if (im->d) (void)hipFree(im->d);
    *im = RtImage();
}

static RtPostState *post_state(RtContext *ctx)
{
    if (!ctx->post) ctx->post = new (std::nothrow) RtPostState();
    return ctx->post;
}

void rt_free_post(RtContext *ctx)
{
    RtPostState *ps = ctx->post;
    if (!ps) return;
    image_free(&ps->post);
    image_free(&ps->user);
    image_free(&ps->staged);
    image_free(&ps->lut);
    image_free(&ps->thr);
    for (RtImage &im : ps->downs) image_free(&im);
    for (RtImage &im : ps->ups) image_free(&im);
    if (ps->dWeights) (void)hipFree(ps->dWeights);
    if (ps->dSrgbTable) (void)hipFree(ps->dSrgbTable);
    if (ps->dRgb) (void)hipFree(ps->dRgb);
    delete ps;
    ctx->post = nullptr;
}

// The frame behind an RT_IMAGE_* selector as a whole-frame device image.  The path
// tracer's own images live in tile-local order and are expanded into `staged`.
static int resolve_image(RtContext *ctx, int which, RtImage *out)
{
    RtPostState *ps = post_state(ctx);
    if (!ps) return rt_fail(ctx, RT_E_NOMEM, "out of host memory");
    if (which == RT_IMAGE_POSTPROCESS || which == RT_IMAGE_USER) {
        const RtImage &im = which == RT_IMAGE_USER ? ps->user : ps->post;
        if (!im.d || im.w <= 0) return rt_fail(ctx, RT_E_STATE, which == RT_IMAGE_USER ? "no image has been uploaded with rt_image_upload" : "no post-process stage has run yet");
        *out = im;
        return RT_OK;
    }
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    const float4 *local = rt_image_ptr(ctx, which);
    if (!local) return rt_fail(ctx, RT_E_INVALID, "bad image selector %d", which);
    int rc = image_reserve(ctx, &ps->staged, ctx->params.width, ctx->params.height);
    if (rc) return rc;
    RT_HIP(ctx, hipMemsetAsync(ps->staged.d, 0, (size_t)ps->staged.w * ps->staged.h * sizeof(float4), ctx->stream));
    if ((rc = rt_expand_local_impl(ctx, local, ps->staged.d))) return rc;
    *out = ps->staged;
    return RT_OK;
}

static int finish(RtContext *ctx)
{
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_timer_resolve(ctx);
    return RT_OK;
}

// ------------------------------------------------------------------ device helpers
__device__ inline float4 tex_bilinear(const float4 *__restrict__ t, int w, int h, float u, float v, int wrap)
{
    const RtTexAxis ax = rt_tex_bilinear_axis(u, w, wrap);
    const RtTexAxis ay = rt_tex_bilinear_axis(v, h, wrap);
    const float4 t00 = t[(size_t)ay.i0 * w + ax.i0], t10 = t[(size_t)ay.i0 * w + ax.i1];
    const float4 t01 = t[(size_t)ay.i1 * w + ax.i0], t11 = t[(size_t)ay.i1 * w + ax.i1];
    return make_float4(rt_bilerp(t00.x, t10.x, t01.x, t11.x, ax.f, ay.f), rt_bilerp(t00.y, t10.y, t01.y, t11.y, ax.f, ay.f),
                       rt_bilerp(t00.z, t10.z, t01.z, t11.z, ax.f, ay.f), rt_bilerp(t00.w, t10.w, t01.w, t11.w, ax.f, ay.f));
}

__device__ inline float4 tex_point(const float4 *__restrict__ t, int w, int h, float u, float v, int wrap)
{
    return t[(size_t)rt_tex_point_axis(v, h, wrap) * w + rt_tex_point_axis(u, w, wrap)];
}

// GaussNxN (DownSample.shader:56-75, UpSample.shader:57-76): i outer, j inner, weights summed as they come
__device__ inline rt_f3 gauss_nxn(const float4 *__restrict__ tex, int w, int h, float u, float v, int r, float sx, float sy,
                                  const float *__restrict__ wt, int wrap)
{
    rt_f3 color = rt_splat3(0.0f);
    float weight = 0.0f;
    const int side = 2 * r + 1;
    for (int i = -r; i <= r; ++i) {
        const float cu = u + (float)i * sx;
        for (int j = -r; j <= r; ++j) {
            const float wgt = wt[(i + r) * side + (j + r)];
            const float cv = v + (float)j * sy;
            const float4 t = tex_bilinear(tex, w, h, cu, cv, wrap);
            color.x = color.x + t.x * wgt;
            color.y = color.y + t.y * wgt;
            color.z = color.z + t.z * wgt;
            weight = weight + wgt;
        }
    }
    return rt_divs3(color, weight);
}

// ------------------------------------------------------------------ kernels
#define POST_THREADS 256

__global__ __launch_bounds__(POST_THREADS) void k_post_blit(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ColorCorrection.shader:46-70
__global__ __launch_bounds__(POST_THREADS) void k_post_color_correction(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n,
                                                                        float brightness, float saturation, float contrast)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 col = src[i];
    const rt_f3 rgb = rt_mk3(col.x, col.y, col.z);
    rt_f3 color = rt_scale3(rgb, brightness);
    const float minGray = rt_dot3(rt_mk3(0.2125f, 0.7154f, 0.0721f), rgb);
    color = rt_lerp3(rt_splat3(minGray), color, saturation);
    color = rt_lerp3(rt_splat3(0.5f), color, contrast);
    dst[i] = make_float4(color.x, color.y, color.z, col.w);
}

// ColorGrading.shader:46-68
__global__ __launch_bounds__(POST_THREADS) void k_post_color_grading(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n,
                                                                     const float4 *__restrict__ lut, int lw, int lh)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 c = src[i];
    const float r = rt_unity_linear_to_gamma1(c.x), g = rt_unity_linear_to_gamma1(c.y), b = rt_unity_linear_to_gamma1(c.z);
    const float blockSize = 32.0f;
    const float leftBlockIndex = rt_floor(b * blockSize);
    const float rightBlockIndex = rt_min(leftBlockIndex + 1.0f, blockSize - 1.0f);
    const float threshold = (blockSize - 1.0f) / blockSize;
    const float inBlock = r * threshold / blockSize;
    const float vv = g * threshold;
    const float4 col0 = tex_point(lut, lw, lh, leftBlockIndex / blockSize + inBlock, vv, RT_WRAP_REPEAT);
    const float4 col1 = tex_point(lut, lw, lh, rightBlockIndex / blockSize + inBlock, vv, RT_WRAP_REPEAT);
    const float f = rt_frac(b * blockSize);
    dst[i] = make_float4(rt_unity_gamma_to_linear1(rt_lerp(col0.x, col1.x, f)), rt_unity_gamma_to_linear1(rt_lerp(col0.y, col1.y, f)),
                         rt_unity_gamma_to_linear1(rt_lerp(col0.z, col1.z, f)), c.w);
}

// Threshold.shader:45-54
__global__ __launch_bounds__(POST_THREADS) void k_post_threshold(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n, float threshold)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 col = src[i];
    const float lum = rt_dot3(rt_mk3(0.2126f, 0.7152f, 0.0722f), rt_mk3(col.x, col.y, col.z));
    dst[i] = lum > threshold ? col : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

// DownSample.shader:77-84 drawn into a dw x dh target
__global__ __launch_bounds__(POST_THREADS) void k_post_downsample(const float4 *__restrict__ src, int sw, int sh, float4 *__restrict__ dst, int dw, int dh,
                                                                  const float *__restrict__ wt, int r, float sx, float sy, int wrap)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= (uint32_t)dw * (uint32_t)dh) return;
    const int px = (int)(i % (uint32_t)dw), py = (int)(i / (uint32_t)dw);
    const rt_f3 c = gauss_nxn(src, sw, sh, rt_blit_uv(px, dw), rt_blit_uv(py, dh), r, sx, sy, wt, wrap);
    dst[i] = make_float4(c.x, c.y, c.z, 0.0f);
}

// UpSample.shader:78-91: target has the size of `cur`
__global__ __launch_bounds__(POST_THREADS) void k_post_upsample(const float4 *__restrict__ cur, int cw, int ch, const float4 *__restrict__ prev, int pw, int ph,
                                                                float4 *__restrict__ dst, const float *__restrict__ wt, int r, float sx, float sy, int wrap)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= (uint32_t)cw * (uint32_t)ch) return;
    const int px = (int)(i % (uint32_t)cw), py = (int)(i / (uint32_t)cw);
    const float u = rt_blit_uv(px, cw), v = rt_blit_uv(py, ch);
    const rt_f3 p = gauss_nxn(prev, pw, ph, u, v, r, 0.5f * sx, 0.5f * sy, wt, wrap);
    const rt_f3 c = gauss_nxn(cur, cw, ch, u, v, r, sx, sy, wt, wrap);
    dst[i] = make_float4(p.x + c.x, p.y + c.y, p.z + c.z, 0.0f);
}

// Bloom.shader:58-71
__global__ __launch_bounds__(POST_THREADS) void k_post_bloom_compose(const float4 *__restrict__ src, int w, int h, const float4 *__restrict__ bloomTex, int bw, int bh,
                                                                     float4 *__restrict__ dst, float intensity, int wrap)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    const int px = (int)(i % (uint32_t)w), py = (int)(i / (uint32_t)w);
    float4 col = src[i];
    const float4 t = tex_bilinear(bloomTex, bw, bh, rt_blit_uv(px, w), rt_blit_uv(py, h), wrap);
    const float g = 1.0f / 2.2f;
    const float bx = rt_saturate(rt_pow(rt_aces1(t.x * intensity, 1.0f), g));
    const float by = rt_saturate(rt_pow(rt_aces1(t.y * intensity, 1.0f), g));
    const float bz = rt_saturate(rt_pow(rt_aces1(t.z * intensity, 1.0f), g));
    col.x += bx;
    col.y += by;
    col.z += bz;
    dst[i] = col;
}

// SaveTexture.cs:25-36: linear float -> 8 bit, table look-up (LinearToSRGB of k/255, 8 bit again)
__global__ __launch_bounds__(POST_THREADS) void k_post_rgb24(const float4 *__restrict__ src, uint32_t n, const uint32_t *__restrict__ table, uint8_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * POST_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 c = src[i];
    dst[(size_t)i * 3 + 0] = (uint8_t)table[rt_unorm8(c.x)];
    dst[(size_t)i * 3 + 1] = (uint8_t)table[rt_unorm8(c.y)];
    dst[(size_t)i * 3 + 2] = (uint8_t)table[rt_unorm8(c.z)];
}

static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + POST_THREADS - 1) / POST_THREADS)); }

// ------------------------------------------------------------------ host side
// SaveTexture.LinearToSRGB (SaveTexture.cs:76-85): Mathf.Pow goes through double, Mathf.Lerp is a + (b - a) * t.
void rt_save_srgb_table(uint32_t table[256])
{
    for (int k = 0; k < 256; ++k) {
        const float v = (float)k / 255.0f;
        const float a = (float)pow((double)v, (double)(1.0f / 2.4f)) * 1.055f - 0.055f;
        const float b = v * 12.92f;
        const float t = v < 0.0031308f ? 1.0f : 0.0f;
        table[k] = rt_unorm8(a + (b - a) * t);
    }
}

static int upload_weights(RtContext *ctx, RtPostState *ps, int slot, int n, float sigma)
{
    if (!ps->dWeights) RT_HIP(ctx, hipMalloc((void **)&ps->dWeights, 2 * 225 * sizeof(float)));
    float w[225];
    const int r = n / 2, side = 2 * r + 1;
    for (int i = -r; i <= r; ++i)
        for (int j = -r; j <= r; ++j) w[(i + r) * side + (j + r)] = rt_gauss_weight2d((float)i, (float)j, sigma);
    RT_HIP(ctx, hipMemcpyAsync(ps->dWeights + slot * 225, w, (size_t)side * side * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // w is on the stack
    return RT_OK;
}

int rt_post_readback_impl(RtContext *ctx, int which, float *dst_rgba)
{
    RtImage im;
    int rc = resolve_image(ctx, which, &im);
    if (rc) return rc;
    RT_HIP(ctx, hipMemcpyAsync(dst_rgba, im.d, (size_t)im.w * im.h * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

#define POST_ENTER(ctx)                                                                  \
    if (!(ctx)) return RT_E_INVALID;                                                     \
    if (hipSetDevice((ctx)->device) != hipSuccess) return rt_fail((ctx), RT_E_DEVICE, "hipSetDevice failed")

extern "C" {

int rt_image_upload(RtContext *ctx, const float *rgba, int32_t width, int32_t height)
{
    POST_ENTER(ctx);
    if (!rgba || width <= 0 || height <= 0 || width > 32768 || height > 32768) return rt_fail(ctx, RT_E_INVALID, "bad image %d x %d", width, height);
    RtPostState *ps = post_state(ctx);
    if (!ps) return rt_fail(ctx, RT_E_NOMEM, "out of host memory");
    int rc = image_reserve(ctx, &ps->user, width, height);
    if (rc) return rc;
    RT_HIP(ctx, hipMemcpyAsync(ps->user.d, rgba, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_image_size(RtContext *ctx, int which, int32_t *width, int32_t *height)
{
    if (!ctx) return RT_E_INVALID;
    int w = 0, h = 0;
    if (which == RT_IMAGE_POSTPROCESS || which == RT_IMAGE_USER) {
        const RtPostState *ps = ctx->post;
        const RtImage *im = !ps ? nullptr : (which == RT_IMAGE_USER ? &ps->user : &ps->post);
        if (!im || !im->d) return rt_fail(ctx, RT_E_STATE, "image %d does not exist yet", which);
        w = im->w;
        h = im->h;
    } else if (which >= RT_IMAGE_RT && which <= RT_IMAGE_SHADOWMAP) {
        if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
        w = ctx->params.width;
        h = ctx->params.height;
    } else {
        return rt_fail(ctx, RT_E_INVALID, "bad image selector %d", which);
    }
    if (width) *width = w;
    if (height) *height = h;
    return RT_OK;
}

int rt_post_blit(RtContext *ctx, int src)
{
    POST_ENTER(ctx);
    RtImage s;
    int rc = resolve_image(ctx, src, &s);
    if (rc) return rc;
    RtPostState *ps = ctx->post;
    if (src == RT_IMAGE_POSTPROCESS) return RT_OK;
    if ((rc = image_reserve(ctx, &ps->post, s.w, s.h))) return rc;
    const size_t n = (size_t)s.w * s.h;
    rt_timer_begin(ctx, "k_post_blit");
    hipLaunchKernelGGL(k_post_blit, grid_for(n), dim3(POST_THREADS), 0, ctx->stream, s.d, ps->post.d, (uint32_t)n);
    rt_timer_end(ctx);
    return finish(ctx);
}

int rt_post_color_correction(RtContext *ctx, int src, float brightness, float saturation, float contrast)
{
    POST_ENTER(ctx);
    RtImage s;
    int rc = resolve_image(ctx, src, &s);
    if (rc) return rc;
    RtPostState *ps = ctx->post;
    if ((rc = image_reserve(ctx, &ps->post, s.w, s.h))) return rc;
    if (src == RT_IMAGE_POSTPROCESS) s.d = ps->post.d;
    const size_t n = (size_t)s.w * s.h;
    rt_timer_begin(ctx, "k_post_color_correction");
    hipLaunchKernelGGL(k_post_color_correction, grid_for(n), dim3(POST_THREADS), 0, ctx->stream, s.d, ps->post.d, (uint32_t)n, brightness, saturation,
                       contrast);
    rt_timer_end(ctx);
    return finish(ctx);
}

int rt_post_lut_upload(RtContext *ctx, const float *rgba, int32_t width, int32_t height)
{
    POST_ENTER(ctx);
    if (!rgba || width <= 0 || height <= 0 || width > 16384 || height > 16384) return rt_fail(ctx, RT_E_INVALID, "bad LUT %d x %d", width, height);
    RtPostState *ps = post_state(ctx);
    if (!ps) return rt_fail(ctx, RT_E_NOMEM, "out of host memory");
    int rc = image_reserve(ctx, &ps->lut, width, height);
    if (rc) return rc;
    RT_HIP(ctx, hipMemcpyAsync(ps->lut.d, rgba, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_post_color_grading(RtContext *ctx, int src)
{
    POST_ENTER(ctx);
    RtImage s;
    int rc = resolve_image(ctx, src, &s);
    if (rc) return rc;
    RtPostState *ps = ctx->post;
    // a material without a texture samples Unity's default "white" (ColorGrading.shader:6); the host has to give one here
    if (!ps->lut.d) return rt_fail(ctx, RT_E_STATE, "rt_post_lut_upload has not been called");
    if ((rc = image_reserve(ctx, &ps->post, s.w, s.h))) return rc;
    if (src == RT_IMAGE_POSTPROCESS) s.d = ps->post.d;
    const size_t n = (size_t)s.w * s.h;
    rt_timer_begin(ctx, "k_post_color_grading");
    hipLaunchKernelGGL(k_post_color_grading, grid_for(n), dim3(POST_THREADS), 0, ctx->stream, s.d, ps->post.d, (uint32_t)n, ps->lut.d, ps->lut.w, ps->lut.h);
    rt_timer_end(ctx);
    return finish(ctx);
}

int rt_post_bloom(RtContext *ctx, int src, const RtBloomParams *bp)
{
    POST_ENTER(ctx);
    if (!bp) return rt_fail(ctx, RT_E_INVALID, "null bloom parameters");
    const int n = bp->downSampleStep;
    if (n < 2 || n > 16) return rt_fail(ctx, RT_E_INVALID, "downSampleStep %d outside [2,16] (the reference indexes bloomUps[n-2])", n);
    if (bp->downSampleBlurSize < 1 || bp->downSampleBlurSize > 15 || bp->upSampleBlurSize < 1 || bp->upSampleBlurSize > 15)
        return rt_fail(ctx, RT_E_INVALID, "blur size outside [1,15]");
    if (!(bp->downSampleBlurSigma > 0.0f) || !(bp->upSampleBlurSigma > 0.0f)) return rt_fail(ctx, RT_E_INVALID, "blur sigma must be positive");
    if (bp->wrapMode != RT_WRAP_CLAMP && bp->wrapMode != RT_WRAP_REPEAT) return rt_fail(ctx, RT_E_INVALID, "bad wrapMode %d", bp->wrapMode);
    RtImage s;
    int rc = resolve_image(ctx, src, &s);
    if (rc) return rc;
    RtPostState *ps = ctx->post;
    // int w = Screen.width / downSize, downSize = 2, 4, ... (Bloom.cs:56-67)
    if ((s.w >> n) < 1 || (s.h >> n) < 1) return rt_fail(ctx, RT_E_INVALID, "a %d x %d frame is too small for %d down-sample steps", s.w, s.h, n);
    if ((rc = image_reserve(ctx, &ps->post, s.w, s.h))) return rc;
    if (src == RT_IMAGE_POSTPROCESS) s.d = ps->post.d;
    if ((rc = image_reserve(ctx, &ps->thr, s.w, s.h))) return rc;
    if ((int)ps->downs.size() < n) ps->downs.resize(n);
    if ((int)ps->ups.size() < n) ps->ups.resize(n);
    for (int i = 0; i < n; ++i)
        if ((rc = image_reserve(ctx, &ps->downs[i], s.w >> (i + 1), s.h >> (i + 1)))) return rc;
    for (int i = 0; i < n - 1; ++i)
        if ((rc = image_reserve(ctx, &ps->ups[i], ps->downs[n - 2 - i].w, ps->downs[n - 2 - i].h))) return rc;
    if ((rc = upload_weights(ctx, ps, 0, bp->downSampleBlurSize, bp->downSampleBlurSigma))) return rc;
    if ((rc = upload_weights(ctx, ps, 1, bp->upSampleBlurSize, bp->upSampleBlurSigma))) return rc;
    const float sx = bp->mainTexST[0] == 0.0f && bp->mainTexST[1] == 0.0f ? 1.0f : bp->mainTexST[0];
    const float sy = bp->mainTexST[0] == 0.0f && bp->mainTexST[1] == 0.0f ? 1.0f : bp->mainTexST[1];
    const int wrap = bp->wrapMode;
    const float *wDown = ps->dWeights, *wUp = ps->dWeights + 225;
    const int rDown = bp->downSampleBlurSize / 2, rUp = bp->upSampleBlurSize / 2;

    // Step 1 (Bloom.cs:43-49)
    const size_t px = (size_t)s.w * s.h;
    rt_timer_begin(ctx, "k_post_threshold");
    hipLaunchKernelGGL(k_post_threshold, grid_for(px), dim3(POST_THREADS), 0, ctx->stream, s.d, ps->thr.d, (uint32_t)px, bp->luminanceThreshold);
    rt_timer_end(ctx);
    // Step 2 (Bloom.cs:52-72)
    for (int i = 0; i < n; ++i) {
        const RtImage &from = i == 0 ? ps->thr : ps->downs[i - 1];
        const RtImage &to = ps->downs[i];
        rt_timer_begin(ctx, "k_post_downsample");
        hipLaunchKernelGGL(k_post_downsample, grid_for((size_t)to.w * to.h), dim3(POST_THREADS), 0, ctx->stream, from.d, from.w, from.h, to.d, to.w, to.h, wDown,
                           rDown, sx, sy, wrap);
        rt_timer_end(ctx);
    }
    // Step 3 (Bloom.cs:75-100): ups[i] = blur(downs[n-2-i]) + blur(i == 0 ? downs[n-1] : ups[i-1])
    for (int i = 0; i < n - 1; ++i) {
        const RtImage &cur = ps->downs[n - 2 - i];
        const RtImage &prev = i == 0 ? ps->downs[n - 1] : ps->ups[i - 1];
        rt_timer_begin(ctx, "k_post_upsample");
        hipLaunchKernelGGL(k_post_upsample, grid_for((size_t)cur.w * cur.h), dim3(POST_THREADS), 0, ctx->stream, cur.d, cur.w, cur.h, prev.d, prev.w, prev.h,
                           ps->ups[i].d, wUp, rUp, sx, sy, wrap);
        rt_timer_end(ctx);
    }
    // Step 4 (Bloom.cs:103-107)
    const RtImage &bl = ps->ups[n - 2];
    rt_timer_begin(ctx, "k_post_bloom_compose");
    hipLaunchKernelGGL(k_post_bloom_compose, grid_for(px), dim3(POST_THREADS), 0, ctx->stream, s.d, s.w, s.h, bl.d, bl.w, bl.h, ps->post.d, bp->bloomIntensity, wrap);
    rt_timer_end(ctx);
    return finish(ctx);
}

int rt_save_rgb24(RtContext *ctx, int which, uint8_t *dst_rgb)
{
    POST_ENTER(ctx);
    if (!dst_rgb) return rt_fail(ctx, RT_E_INVALID, "null destination");
    RtImage s;
    int rc = resolve_image(ctx, which, &s);
    if (rc) return rc;
    RtPostState *ps = ctx->post;
    if (!ps->dSrgbTable) {
        uint32_t table[256];
        rt_save_srgb_table(table);
        RT_HIP(ctx, hipMalloc((void **)&ps->dSrgbTable, sizeof table));
        RT_HIP(ctx, hipMemcpy(ps->dSrgbTable, table, sizeof table, hipMemcpyHostToDevice));
    }
    const size_t n = (size_t)s.w * s.h;
    if (n * 3 > ps->rgbCap) {
        if (ps->dRgb) (void)hipFree(ps->dRgb);
        ps->dRgb = nullptr;
        ps->rgbCap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ps->dRgb, n * 3));
        ps->rgbCap = n * 3;
    }
    rt_timer_begin(ctx, "k_post_rgb24");
    hipLaunchKernelGGL(k_post_rgb24, grid_for(n), dim3(POST_THREADS), 0, ctx->stream, s.d, (uint32_t)n, ps->dSrgbTable, ps->dRgb);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(dst_rgb, ps->dRgb, n * 3, hipMemcpyDeviceToHost, ctx->stream));
    return finish(ctx);
}

}   // extern "C"
