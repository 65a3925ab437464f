// rt_post_oracle.cpp -- CPU restatement of the reference's post-process chain and of
// SaveTexture's 8-bit conversion.  TEST INFRASTRUCTURE ONLY: nothing the library ships
// links or calls this file; tests, the smoke check and bench.py's CPU leg use it as
// the checker.
//
// Structure follows the reference: a fragment function per shader, drawn over every
// destination pixel by Blit() the way Graphics.Blit draws a full-screen quad.
//
// Parity unpinned for exact values: the reference holds no test or known-answer vector for this
// chain, the samplers are fixed-function hardware and LinearToGammaSpace / GammaToLinearSpace
// live in Unity's UnityCG.cginc (Unity 2021.3.1f1 built-in shaders, not in the reference tree;
// restated in include/rt_post.h from their published text).  What IS anchored on the
// reference's own data: grading through its NeutralLUT_32.png must be the identity up to the
// LUT's resolution (tests/test_post_cpu.py), and Assets/Outputs/bloom1.png, published of
// Assets/Scenes/Chapter7_5_1.unity with that scene's Bloom component on, is matched to a median
// of 15 / 255 per 16 x 16 block with the stage and 46 without (tests/test_gpu_scale.py).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../include/rt_post.h"
#include "../include/rtcore.h"

namespace {

struct Float4 {
    float x, y, z, w;
};

// a sampler2D bound to an image in memory (row 0 = v 0)
struct Texture {
    const Float4 *texels;
    int width, height;
    int wrap;        // RT_WRAP_*
    bool bilinear;   // FilterMode.Bilinear, else Point
};

Float4 tex2D(const Texture &t, float u, float v)
{
    if (!t.bilinear) {
        const int x = rt_tex_point_axis(u, t.width, t.wrap), y = rt_tex_point_axis(v, t.height, t.wrap);
        return t.texels[(size_t)y * t.width + x];
    }
    const RtTexAxis ax = rt_tex_bilinear_axis(u, t.width, t.wrap);
    const RtTexAxis ay = rt_tex_bilinear_axis(v, t.height, t.wrap);
    const Float4 &a = t.texels[(size_t)ay.i0 * t.width + ax.i0], &b = t.texels[(size_t)ay.i0 * t.width + ax.i1];
    const Float4 &c = t.texels[(size_t)ay.i1 * t.width + ax.i0], &d = t.texels[(size_t)ay.i1 * t.width + ax.i1];
    return Float4{rt_bilerp(a.x, b.x, c.x, d.x, ax.f, ay.f), rt_bilerp(a.y, b.y, c.y, d.y, ax.f, ay.f), rt_bilerp(a.z, b.z, c.z, d.z, ax.f, ay.f),
                  rt_bilerp(a.w, b.w, c.w, d.w, ax.f, ay.f)};
}

// Graphics.Blit(source, destination, material): frag(uv) for the centre of every destination pixel
void Blit(std::vector<Float4> &destination, int dw, int dh, const std::function<Float4(float, float)> &frag)
{
    destination.resize((size_t)dw * dh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) destination[(size_t)y * dw + x] = frag(rt_blit_uv(x, dw), rt_blit_uv(y, dh));
}

// ColorCorrection.shader:46-70
Float4 ColorCorrectionFrag(const Texture &_MainTex, float u, float v, float _Brightness, float _Saturation, float _Contrast)
{
    const Float4 col = tex2D(_MainTex, u, v);
    float color[3] = {col.x * _Brightness, col.y * _Brightness, col.z * _Brightness};
    const float minGray = (0.2125f * col.x + 0.7154f * col.y) + 0.0721f * col.z;
    for (float &c : color) c = rt_lerp(minGray, c, _Saturation);
    for (float &c : color) c = rt_lerp(0.5f, c, _Contrast);
    return Float4{color[0], color[1], color[2], col.w};
}

// ColorGrading.shader:46-68
Float4 ColorGradingFrag(const Texture &_MainTex, const Texture &_LutTex, float u, float v)
{
    Float4 col = tex2D(_MainTex, u, v);
    col.x = rt_unity_linear_to_gamma1(col.x);
    col.y = rt_unity_linear_to_gamma1(col.y);
    col.z = rt_unity_linear_to_gamma1(col.z);
    const float blockSize = 32.0f;
    const float leftBlockIndex = floorf(col.z * blockSize);
    const float rightBlockIndex = rt_min(leftBlockIndex + 1.0f, blockSize - 1.0f);
    const float threshold = (blockSize - 1.0f) / blockSize;
    const float uvLeftX = leftBlockIndex / blockSize + col.x * threshold / blockSize;
    const float uvRightX = rightBlockIndex / blockSize + col.x * threshold / blockSize;
    const float uvY = col.y * threshold;
    const Float4 col0 = tex2D(_LutTex, uvLeftX, uvY);
    const Float4 col1 = tex2D(_LutTex, uvRightX, uvY);
    const float f = rt_frac(col.z * blockSize);
    col.x = rt_unity_gamma_to_linear1(rt_lerp(col0.x, col1.x, f));
    col.y = rt_unity_gamma_to_linear1(rt_lerp(col0.y, col1.y, f));
    col.z = rt_unity_gamma_to_linear1(rt_lerp(col0.z, col1.z, f));
    return col;
}

// Threshold.shader:45-54
Float4 ThresholdFrag(const Texture &_MainTex, float u, float v, float _LuminanceThreshold)
{
    const Float4 col = tex2D(_MainTex, u, v);
    const float lum = (0.2126f * col.x + 0.7152f * col.y) + 0.0722f * col.z;
    if (lum > _LuminanceThreshold) return col;
    return Float4{0.0f, 0.0f, 0.0f, 1.0f};
}

// GaussNxN (DownSample.shader:56-75 = UpSample.shader:57-76)
void GaussNxN(const Texture &tex, float u, float v, int n, float strideX, float strideY, float sigma, float out[3])
{
    float color[3] = {0.0f, 0.0f, 0.0f};
    const int r = n / 2;
    float weight = 0.0f;
    for (int i = -r; i <= r; i++) {
        for (int j = -r; j <= r; j++) {
            const float w = rt_gauss_weight2d((float)i, (float)j, sigma);
            const Float4 t = tex2D(tex, u + (float)i * strideX, v + (float)j * strideY);
            color[0] = color[0] + t.x * w;
            color[1] = color[1] + t.y * w;
            color[2] = color[2] + t.z * w;
            weight = weight + w;
        }
    }
    for (int k = 0; k < 3; ++k) out[k] = color[k] / weight;
}

// Bloom.shader:46-71
Float4 BloomFrag(const Texture &_MainTex, const Texture &_BloomTex, float u, float v, float _BloomIntensity)
{
    Float4 col = tex2D(_MainTex, u, v);
    const Float4 t = tex2D(_BloomTex, u, v);
    float bloom[3] = {t.x * _BloomIntensity, t.y * _BloomIntensity, t.z * _BloomIntensity};
    const float g = 1.0f / 2.2f;
    for (float &b : bloom) b = rt_saturate(rt_pow(rt_aces1(b, 1.0f), g));
    col.x += bloom[0];
    col.y += bloom[1];
    col.z += bloom[2];
    return col;
}

Texture RenderTextureOf(const std::vector<Float4> &img, int w, int h, int wrap) { return Texture{img.data(), w, h, wrap, true}; }

}   // namespace

extern "C" {

void orc_post_color_correction(const float *src, float *dst, int w, int h, float brightness, float saturation, float contrast)
{
    const Texture source{(const Float4 *)src, w, h, RT_WRAP_CLAMP, true};
    std::vector<Float4> destination;
    Blit(destination, w, h, [&](float u, float v) { return ColorCorrectionFrag(source, u, v, brightness, saturation, contrast); });
    memcpy(dst, destination.data(), destination.size() * sizeof(Float4));
}

void orc_post_color_grading(const float *src, float *dst, int w, int h, const float *lut, int lw, int lh)
{
    const Texture source{(const Float4 *)src, w, h, RT_WRAP_CLAMP, true};
    const Texture lutTex{(const Float4 *)lut, lw, lh, RT_WRAP_REPEAT, false};
    std::vector<Float4> destination;
    Blit(destination, w, h, [&](float u, float v) { return ColorGradingFrag(source, lutTex, u, v); });
    memcpy(dst, destination.data(), destination.size() * sizeof(Float4));
}

// Bloom.RenderImage (Bloom.cs:41-115).  Returns 0, or -1 when a pyramid level would be empty.
int orc_post_bloom(const float *src, float *dst, int w, int h, const RtBloomParams *bp)
{
    const int n = bp->downSampleStep;
    if (n < 2) return -1;
    const bool stDefault = bp->mainTexST[0] == 0.0f && bp->mainTexST[1] == 0.0f;
    const float stX = stDefault ? 1.0f : bp->mainTexST[0], stY = stDefault ? 1.0f : bp->mainTexST[1];
    const int wrap = bp->wrapMode;
    const Texture source{(const Float4 *)src, w, h, wrap, true};

    // Step 1
    std::vector<Float4> thresholdRT;
    Blit(thresholdRT, w, h, [&](float u, float v) { return ThresholdFrag(source, u, v, bp->luminanceThreshold); });

    // Step 2
    std::vector<std::vector<Float4>> bloomDowns(n), bloomUps(n);
    std::vector<int> dw(n), dh(n);
    int downSize = 2;
    for (int i = 0; i < n; ++i) {
        dw[i] = w / downSize;
        dh[i] = h / downSize;
        if (dw[i] < 1 || dh[i] < 1) return -1;
        downSize *= 2;
    }
    for (int i = 0; i < n; ++i) {
        const Texture from = i == 0 ? RenderTextureOf(thresholdRT, w, h, wrap) : RenderTextureOf(bloomDowns[i - 1], dw[i - 1], dh[i - 1], wrap);
        Blit(bloomDowns[i], dw[i], dh[i], [&](float u, float v) {
            float c[3];
            GaussNxN(from, u, v, bp->downSampleBlurSize, stX, stY, bp->downSampleBlurSigma, c);
            return Float4{c[0], c[1], c[2], 0.0f};
        });
    }

    // Step 3
    for (int i = 0; i < n - 1; ++i) {
        const int cw = dw[n - 2 - i], ch = dh[n - 2 - i];
        const Texture curMip = RenderTextureOf(bloomDowns[n - 2 - i], cw, ch, wrap);
        const Texture prevMip = i == 0 ? RenderTextureOf(bloomDowns[n - 1], dw[n - 1], dh[n - 1], wrap)
                                       : RenderTextureOf(bloomUps[i - 1], dw[n - 1 - i], dh[n - 1 - i], wrap);
        Blit(bloomUps[i], cw, ch, [&](float u, float v) {
            float p[3], c[3];
            GaussNxN(prevMip, u, v, bp->upSampleBlurSize, 0.5f * stX, 0.5f * stY, bp->upSampleBlurSigma, p);
            GaussNxN(curMip, u, v, bp->upSampleBlurSize, stX, stY, bp->upSampleBlurSigma, c);
            return Float4{p[0] + c[0], p[1] + c[1], p[2] + c[2], 0.0f};
        });
    }

    // Step 4
    const Texture bloomTex = RenderTextureOf(bloomUps[n - 2], dw[0], dh[0], wrap);
    std::vector<Float4> destination;
    Blit(destination, w, h, [&](float u, float v) { return BloomFrag(source, bloomTex, u, v, bp->bloomIntensity); });
    memcpy(dst, destination.data(), destination.size() * sizeof(Float4));
    return 0;
}

// SaveTexture.SavePNG up to EncodeToPNG (SaveTexture.cs:23-38) and LinearToSRGB (:59-90)
void orc_save_rgb24(const float *src, int w, int h, uint8_t *dst)
{
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int c = 0; c < 3; ++c) {
            const uint32_t q = rt_unorm8(src[i * 4 + c]);         // ReadPixels into RGB24
            const float v = (float)q / 255.0f;                    // GetPixels
            const float a = (float)pow((double)v, (double)(1.0f / 2.4f)) * 1.055f - 0.055f;   // Mathf.Pow
            const float b = v * 12.92f;
            const float t = v < 0.0031308f ? 1.0f : 0.0f;         // LessThan
            dst[i * 3 + c] = (uint8_t)rt_unorm8(a + (b - a) * t); // Mathf.Lerp, SetPixels
        }
}

}   // extern "C"
