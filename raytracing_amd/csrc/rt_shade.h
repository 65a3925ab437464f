// rt_shade.h -- device-side Monte Carlo shading for one path vertex.
//
// Replaces Shade() and everything under it in the reference:
//   Assets/Shaders/RayTracing/Shade.cginc:30-103
//   Assets/Shaders/RayTracing/ImportanceSampling.cginc:42-395
//   Assets/Shaders/RayTracing/BRDF.cginc:4-204
// One call = one bounce of one path; every Rand() draw of that bounce happens
// inside this call in the reference's order, so the per-pixel random stream is
// the same as in the one-thread-per-pixel loop of RayTracing.compute:70-88.
// Arithmetic uses only include/rt_math.h primitives, in a fixed association
// order, and must be compiled with -ffp-contract=off.
#ifndef RT_SHADE_H_
#define RT_SHADE_H_

#include "rt_internal.h"

typedef rt_f3 f3;

struct DRay {
    f3 origin, direction, energy;
};

struct DHit {
    f3 position;
    float distance;
    f3 normal;
    f3 albedo;
    float metallic, smoothness, transparent;
    f3 emissionColor;
    uint32_t materialType;
    float ior;
};

struct DShadeCtx {
    const RtSceneView *sv;
    const RtParams *P;
    rt_rng rng;
};

#define RTD __device__ __forceinline__
#define RTDN __device__ __noinline__

RTD f3 ld3(const float *p) { return rt_mk3(p[0], p[1], p[2]); }
RTD float d_rand(DShadeCtx &c) { return rt_rand(&c.rng); }

#define D_DIELECTRIC_SPEC 0.04f
#define D_ONE_MINUS_DIELECTRIC_SPEC (1.0f - 0.04f)
#define D_TWO_PI (2.0f * RT_PI)

RTD float d_pow5(float x) { return x * x * x * x * x; }

// BRDF.cginc:141-181 with its callees (:32-139) inlined in call order.
RTD void d_brdf(const RtParams &P, f3 viewDir, f3 halfDir, f3 lightDir, f3 albedo, f3 normal, float metallic,
                float perceptualRoughness, float roughness, float diffuseRatio, float specularRoatio, f3 &func, float &pdf)
{
    // DiffuseAndSpecularFromMetallic
    const f3 specColor = rt_lerp3(rt_splat3(D_DIELECTRIC_SPEC), albedo, metallic);
    const float oneMinusReflectivity = D_ONE_MINUS_DIELECTRIC_SPEC - metallic * D_ONE_MINUS_DIELECTRIC_SPEC;
    const f3 diffuse = rt_scale3(albedo, oneMinusReflectivity);
    // DiffuseBRDF
    float diffusePdf;
    f3 diffuseBRDF;
    {
        const float nv = rt_saturate(rt_dot3(normal, viewDir));
        const float nl = rt_saturate(rt_dot3(normal, lightDir));
        const float lh = rt_saturate(rt_dot3(lightDir, halfDir));
        const float fd90 = 0.5f + 2 * lh * lh * perceptualRoughness;
        const float lightScatter = (1 + (fd90 - 1) * d_pow5(1 - nl));
        const float viewScatter = (1 + (fd90 - 1) * d_pow5(1 - nv));
        const float diffuseTerm = (lightScatter * viewScatter) * nl;
        diffuseBRDF = rt_divs3(rt_scale3(diffuse, diffuseTerm), RT_PI);
        diffusePdf = nl / RT_PI;
    }
    // SpecularBRDF
    f3 F, specularBRDF;
    float specularPdf;
    {
        const float nv = rt_abs(rt_dot3(normal, viewDir));
        const float nl = rt_saturate(rt_dot3(normal, lightDir));
        const float nh = rt_saturate(rt_dot3(normal, halfDir));
        const float hv = rt_saturate(rt_dot3(halfDir, viewDir));
        const float a2 = roughness * roughness;
        const float dd = (nh * a2 - nh) * nh + 1.0f;
        const float D = a2 / (RT_PI * (dd * dd + 1e-7f));
        const float t = d_pow5(1 - hv);
        F = rt_add3(specColor, rt_scale3(rt_sub3(rt_splat3(1.0f), specColor), t));
        // GeometrySmith
        const float NdotV = rt_saturate(rt_dot3(normal, viewDir));
        const float NdotL = rt_saturate(rt_dot3(normal, lightDir));
        const float r = (roughness + 1.0f);
        const float k = (r * r) / 8.0f;
        const float ggx2 = NdotV / (NdotV * (1.0f - k) + k);
        const float ggx1 = NdotL / (NdotL * (1.0f - k) + k);
        const float G = ggx1 * ggx2;
        const f3 nominator = rt_scale3(F, D * G);
        const float denominator = 4.0f * nv * nl + 0.001f;
        specularBRDF = rt_divs3(nominator, denominator);
        specularPdf = D * nh / (4.0f * hv);
    }
    const f3 lc = rt_scale3(rt_mk3(P.lightColor[0], P.lightColor[1], P.lightColor[2]), P.lightColor[3]);
    f3 kD = rt_sub3(rt_splat3(1.0f), F);
    kD = rt_scale3(kD, 1.0f - metallic);
    const f3 a = rt_divs3(rt_mul3(rt_mul3(diffuseBRDF, kD), lc), 2.0f);
    const f3 b = rt_mul3(specularBRDF, lc);
    func = rt_scale3(rt_add3(a, b), rt_saturate(rt_dot3(normal, lightDir)));
    pdf = diffusePdf * diffuseRatio + specularPdf * specularRoatio;
}

// BRDF.cginc:184-204
RTD float d_fresnel_reflect_amount(float n1, float n2, f3 normal, f3 incident, float f0, float f90)
{
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -rt_dot3(normal, incident);
    if (n1 > n2) {
        const float n = n1 / n2;
        const float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return f90;
        cosX = rt_sqrt(1.0f - sinT2);
    }
    const float x = 1.0f - cosX;
    const float ret = r0 + (1.0f - r0) * x * x * x * x * x;
    return rt_lerp(f0, f90, ret);
}

// ImportanceSampling.cginc:42-60
RTD f3 d_tangent2world(float theta, float phi, f3 direction)
{
    float st, ct, sp, cp;
    rt_sincos(theta, &st, &ct);
    rt_sincos(phi, &sp, &cp);
    const f3 l = rt_mk3(cp * st, sp * st, ct);
    f3 helper = rt_mk3(1, 0, 0);
    if (rt_abs(direction.x) > 0.99f) helper = rt_mk3(0, 0, 1);
    const f3 T = rt_normalize3(rt_cross3(direction, helper));
    const f3 B = rt_normalize3(rt_cross3(direction, T));
    const f3 N = direction;
    return rt_normalize3(rt_mk3((l.x * T.x + l.y * B.x) + l.z * N.x, (l.x * T.y + l.y * B.y) + l.z * N.y,
                                (l.x * T.z + l.y * B.z) + l.z * N.z));
}

RTD f3 d_offset_origin(const DHit &hit) { return rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS)); }

// ImportanceSampling.cginc:62-99
RTD f3 d_uniform_sampling(DShadeCtx &c, const DHit &hit, DRay &ray)
{
    const float theta = rt_acos(1 - d_rand(c));
    const float phi = D_TWO_PI * d_rand(c);
    ray.origin = d_offset_origin(hit);
    ray.direction = d_tangent2world(theta, phi, hit.normal);
    const float pdf = 1.0f / D_TWO_PI;
    const f3 fr = rt_divs3(hit.albedo, RT_PI);
    return rt_scale3(rt_divs3(fr, pdf), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

RTD f3 d_cosine_sampling(DShadeCtx &c, const DHit &hit, DRay &ray)
{
    const float theta = rt_acos(rt_sqrt(1 - d_rand(c)));
    const float phi = D_TWO_PI * d_rand(c);
    ray.origin = d_offset_origin(hit);
    ray.direction = d_tangent2world(theta, phi, hit.normal);
    return hit.albedo;
}

// ImportanceSampling.cginc:103-118
RTD void d_sphere_light(DShadeCtx &c, const DHit &hit, DRay &ray, f3 lpos, float lradius, f3 &func, float &pdf)
{
    const f3 dir = rt_sub3(lpos, hit.position);
    const float maxCos = rt_sqrt(1 - rt_pow(lradius / rt_length3(dir), 2.0f));
    const float theta = rt_acos(d_rand(c) * (maxCos - 1) + 1);
    const float phi = D_TWO_PI * d_rand(c);
    const f3 direction = rt_normalize3(dir);
    ray.origin = d_offset_origin(hit);
    ray.direction = d_tangent2world(theta, phi, direction);
    pdf = 1.0f / (D_TWO_PI * (1 - maxCos));
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

// ImportanceSampling.cginc:147-160.  Defined by the reference but never called:
// _LightImportanceSampling (:163-181) has no DISC_LIGHT branch, so uploaded disc lights
// light nothing.  Kept callable for the numeric self-check (rt_debug_eval 8).
RTD void d_disc_light(DShadeCtx &c, const DHit &hit, DRay &ray, f3 lpos, float lradius, f3 &func, float &pdf)
{
    const float theta = rt_sqrt(d_rand(c) * lradius);
    const float phi = D_TWO_PI * d_rand(c);
    const f3 direction = rt_normalize3(rt_sub3(lpos, hit.position));
    ray.origin = d_offset_origin(hit);
    ray.direction = d_tangent2world(theta, phi, direction);
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
    pdf = 1 / (RT_PI * lradius * lradius);
}

// ImportanceSampling.cginc:122-143
RTD void d_area_light(DShadeCtx &c, const DHit &hit, DRay &ray, f3 lpos, f3 lnormal, f3 lup, float sx, float sy, f3 &func,
                      float &pdf)
{
    const float x = (d_rand(c) * 2 - 1) * sx / 2;
    const float z = (d_rand(c) * 2 - 1) * sy / 2;
    const f3 bin = rt_normalize3(rt_cross3(lnormal, lup));
    // mul(float3(x,0,z), float3x3(binormal, normal, up)) + position
    const f3 pointWS = rt_add3(rt_mk3((x * bin.x + 0.0f * lnormal.x) + z * lup.x, (x * bin.y + 0.0f * lnormal.y) + z * lup.y,
                                      (x * bin.z + 0.0f * lnormal.z) + z * lup.z),
                               lpos);
    ray.origin = d_offset_origin(hit);
    ray.direction = rt_sub3(pointWS, hit.position);
    const float distanceSquard = rt_dot3(ray.direction, ray.direction);
    const float area = sx * sy;
    const float lightCosine = rt_dot3(rt_normalize3(rt_neg3(ray.direction)), lnormal);
    ray.direction = rt_normalize3(ray.direction);
    pdf = distanceSquard / (lightCosine * area);
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

// ImportanceSampling.cginc:163-188.  func/pdf start as 0 / -1 (they are
// unwritten in the shader when no light keyword is on); an index past the end
// of a light buffer reads a zero light, like an out-of-range StructuredBuffer load.
RTD void d_light_sampling(DShadeCtx &c, const DHit &hit, DRay &ray, f3 &func, float &pdf)
{
    const RtSceneView &sv = *c.sv;
    func = rt_splat3(0.0f);
    pdf = -1.0f;
    const bool hasSphere = sv.nSphere > 0, hasArea = sv.nArea > 0;
    bool useSphere = hasSphere, useArea = hasArea;
    if (hasSphere && hasArea) {
        const float roulette = d_rand(c);
        useSphere = roulette > 0.5f;
        useArea = !useSphere;
    }
    if (useSphere) {
        const uint32_t i = rt_ftou(d_rand(c) * (float)(int)sv.nSphere);
        f3 lp = rt_splat3(0.0f);
        float lr = 0.0f;
        if (i < sv.nSphere) { lp = ld3(sv.sphereLights[i].position); lr = sv.sphereLights[i].radius; }
        d_sphere_light(c, hit, ray, lp, lr, func, pdf);
    }
    if (useArea) {
        const uint32_t i = rt_ftou(d_rand(c) * (float)(int)sv.nArea);
        f3 lp = rt_splat3(0.0f), ln = rt_splat3(0.0f), lu = rt_splat3(0.0f);
        float sx = 0.0f, sy = 0.0f;
        if (i < sv.nArea) {
            const RtAreaLight &L = sv.areaLights[i];
            lp = ld3(L.position); ln = ld3(L.normal); lu = ld3(L.up); sx = L.size[0]; sy = L.size[1];
        }
        d_area_light(c, hit, ray, lp, ln, lu, sx, sy, func, pdf);
    }
    if (rt_dot3(ray.direction, hit.normal) < 0.01f) {
        func = rt_splat3(0.0f);
        pdf = -1;
    }
}

// ImportanceSampling.cginc:204-307
RTD void d_bsdf_sampling(DShadeCtx &c, const DHit &hit, DRay &ray, f3 &func, float &pdf)
{
    const float perceptualRoughness = 1 - hit.smoothness;
    float roughness = perceptualRoughness * perceptualRoughness;
    roughness = rt_max(roughness, 0.002f);
    f3 diffuseReflectionDir, specularReflectionDir;
    {
        const float theta = rt_acos(rt_sqrt(1 - d_rand(c)));
        const float phi = D_TWO_PI * d_rand(c);
        diffuseReflectionDir = d_tangent2world(theta, phi, hit.normal);
    }
    {
        const float e = d_rand(c);
        const float roughness2 = roughness * roughness;
        const float theta = rt_acos(rt_sqrt((1 - e) / (e * (roughness2 - 1) + 1)));
        const float phi = D_TWO_PI * d_rand(c);
        const f3 microfacetNormal = d_tangent2world(theta, phi, hit.normal);
        specularReflectionDir = rt_normalize3(rt_reflect3(ray.direction, microfacetNormal));
    }
    f3 reflectionDir;
    const float diffuseRatio = 0.5f * (1.0f - hit.metallic);
    const float specularRoatio = 1 - diffuseRatio;
    if (d_rand(c) < diffuseRatio) reflectionDir = diffuseReflectionDir;
    else reflectionDir = specularReflectionDir;

    if (hit.materialType == 1 || hit.materialType == 3) {
        const bool fromOutside = rt_dot3(ray.direction, hit.normal) < 0;
        const float etai = 1;
        const float etat = hit.ior;
        const float eta = fromOutside ? etai / etat : etat / etai;
        if (hit.materialType == 3) {
            float specularChance = d_fresnel_reflect_amount(!fromOutside ? etat : etai, fromOutside ? etat : etai,
                                                            ray.direction, hit.normal, D_DIELECTRIC_SPEC, 1.0f);
            specularChance = rt_pow(specularChance, eta * eta * eta * eta);
            const float pick = (d_rand(c) < specularChance) ? 1.0f : 0.0f;
            reflectionDir = rt_lerp3(diffuseReflectionDir, reflectionDir, pick);
        } else if (d_rand(c) > hit.transparent) {
            const f3 N = fromOutside ? hit.normal : rt_neg3(hit.normal);
            const f3 bias = rt_scale3(N, 0.001f);
            ray.origin = rt_sub3(hit.position, bias);
            f3 refractionDir = rt_normalize3(rt_refract3(ray.direction, N, eta));
            refractionDir = rt_normalize3(rt_lerp3(refractionDir, rt_add3(rt_neg3(N), specularReflectionDir), roughness * roughness));
            ray.direction = refractionDir;
            func = rt_splat3(1.0f);
            pdf = 1;
            return;
        }
    }
    const f3 viewDir = rt_normalize3(rt_neg3(ray.direction));
    const f3 halfDir = rt_normalize3(rt_add3(viewDir, reflectionDir));
    d_brdf(*c.P, viewDir, halfDir, reflectionDir, hit.albedo, hit.normal, hit.metallic, perceptualRoughness, roughness,
           diffuseRatio, specularRoatio, func, pdf);
    ray.origin = d_offset_origin(hit);
    ray.direction = reflectionDir;
}

// ImportanceSampling.cginc:309-326
RTD void d_bsdf_eval(DShadeCtx &c, f3 inputDir, f3 outputDir, const DHit &hit, DRay &ray, f3 &func, float &pdf)
{
    const float perceptualRoughness = 1 - hit.smoothness;
    float roughness = perceptualRoughness * perceptualRoughness;
    roughness = rt_max(roughness, 0.002f);
    const float diffuseRatio = 0.5f * (1.0f - hit.metallic);
    const float specularRoatio = 1 - diffuseRatio;
    const f3 viewDir = rt_normalize3(rt_neg3(inputDir));
    const f3 halfDir = rt_normalize3(rt_add3(viewDir, outputDir));
    d_brdf(*c.P, viewDir, halfDir, outputDir, hit.albedo, hit.normal, hit.metallic, perceptualRoughness, roughness,
           diffuseRatio, specularRoatio, func, pdf);
    ray.origin = d_offset_origin(hit);
    ray.direction = outputDir;
}

// Shade.cginc:8-20 dispatch + ImportanceSampling.cginc:190-201,328-395
RTD f3 d_function_bsdf(DShadeCtx &c, const DHit &hit, DRay &ray)
{
    const uint32_t mode = c.P->flags & RT_FLAG_SAMPLING_MASK;
    f3 func;
    float pdf;
    if (mode == RT_SAMPLING_UNIFORM) return d_uniform_sampling(c, hit, ray);
    if (mode == RT_SAMPLING_COSINE) return d_cosine_sampling(c, hit, ray);
    if (mode == RT_SAMPLING_LIGHT) {
        d_light_sampling(c, hit, ray, func, pdf);
        if (pdf > 0) return rt_divs3(func, pdf);
        return rt_splat3(0.0f);
    }
    if (mode == RT_SAMPLING_BSDF) {
        d_bsdf_sampling(c, hit, ray, func, pdf);
        if (pdf > 0) return rt_divs3(func, pdf);
        return rt_splat3(1.0f);
    }
    // MultipleImportanceSampling
    f3 lightFunc;
    float lightPdf;
    const f3 inputDir = ray.direction;
    d_light_sampling(c, hit, ray, lightFunc, lightPdf);
    const f3 outputDir = ray.direction;
    const float coin = d_rand(c);
    const bool facing = rt_dot3(hit.normal, outputDir) > 0.01f;
    if (coin > 0.5f && facing) {
        f3 brdfFunc;
        float brdfPdf;
        d_bsdf_eval(c, inputDir, outputDir, hit, ray, brdfFunc, brdfPdf);
        func = rt_add3(lightFunc, brdfFunc);
        pdf = lightPdf * 0.5f + brdfPdf * 0.5f;
        if (pdf > 0) return rt_divs3(func, pdf);
        return rt_splat3(0.0f);
    }
    ray.direction = inputDir;
    d_bsdf_sampling(c, hit, ray, func, pdf);
    if (pdf > 0) return rt_divs3(func, pdf);
    return rt_splat3(1.0f);
}

// Software cubemap lookup standing in for skyboxCube.SampleLevel(sampler_LinearClamp, dir, 0)
// (Shade.cginc:94): D3D face selection, bilinear inside one face, clamped at
// the face edge; a NaN direction reads texel (0,0) of face +X; unbound reads 0.
RTD f3 d_sample_skybox(const RtSceneView &sv, f3 d)
{
    if (sv.cubeSize == 0) return rt_splat3(0.0f);
    const int size = (int)sv.cubeSize;
    int face = 0;
    float sc = 0, tc = 0, ma = 1;
    const bool nan = d.x != d.x || d.y != d.y || d.z != d.z;
    const float ax = rt_abs(d.x), ay = rt_abs(d.y), az = rt_abs(d.z);
    if (!nan) {
        if (ax >= ay && ax >= az) { face = d.x > 0 ? 0 : 1; ma = ax; sc = d.x > 0 ? -d.z : d.z; tc = -d.y; }
        else if (ay >= az)        { face = d.y > 0 ? 2 : 3; ma = ay; sc = d.x; tc = d.y > 0 ? d.z : -d.z; }
        else                      { face = d.z > 0 ? 4 : 5; ma = az; sc = d.z > 0 ? d.x : -d.x; tc = -d.y; }
    }
    const float4 *img = sv.cube + (size_t)face * size * size;
    if (nan || ma == 0.0f) { const float4 t = img[0]; return rt_mk3(t.x, t.y, t.z); }
    const float u = 0.5f * (sc / ma + 1.0f);
    const float v = 0.5f * (tc / ma + 1.0f);
    const float fx = u * (float)size - 0.5f;
    const float fy = v * (float)size - 0.5f;
    const float x0f = rt_floor(fx), y0f = rt_floor(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), size - 1); x1 = min(max(x1, 0), size - 1);
    y0 = min(max(y0, 0), size - 1); y1 = min(max(y1, 0), size - 1);
    const float4 c00 = img[(size_t)y0 * size + x0], c10 = img[(size_t)y0 * size + x1];
    const float4 c01 = img[(size_t)y1 * size + x0], c11 = img[(size_t)y1 * size + x1];
    const f3 top = rt_lerp3(rt_mk3(c00.x, c00.y, c00.z), rt_mk3(c10.x, c10.y, c10.z), wx);
    const f3 bot = rt_lerp3(rt_mk3(c01.x, c01.y, c01.z), rt_mk3(c11.x, c11.y, c11.z), wx);
    return rt_lerp3(top, bot, wy);
}

// Miss branch of Shade (Shade.cginc:77-102)
RTD f3 d_shade_miss(const RtSceneView &sv, const RtParams &P, f3 direction)
{
    const float alpha = (-P.skyboxRotation) * RT_PI / 180.0f;
    float sina, cosa;
    rt_sincos(alpha, &sina, &cosa);
    const f3 dir = rt_mk3(cosa * direction.x + (-sina) * direction.z, direction.y, sina * direction.x + cosa * direction.z);
    f3 s = d_sample_skybox(sv, dir);
    s = rt_mk3(rt_linear_to_srgb1(s.x), rt_linear_to_srgb1(s.y), rt_linear_to_srgb1(s.z));
    s = rt_scale3(rt_saturate3(s), P.skyboxExposure);
    return rt_mk3(rt_srgb_to_linear1(s.x), rt_srgb_to_linear1(s.y), rt_srgb_to_linear1(s.z));
}

// ---- camera (Header.cginc:110-170) -------------------------------------------
RTD void d_mul44(const float *m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((m[0 * 4 + r] * x + m[1 * 4 + r] * y) + m[2 * 4 + r] * z) + m[3 * 4 + r] * w;
}

RTD void d_camera_ray(const RtParams &P, float uvx, float uvy, f3 &origin, f3 &direction)
{
    float o[4], d[4], dw[4];
    d_mul44(P.camera2World, 0, 0, 0, 1, o);
    d_mul44(P.cameraInverseProjection, uvx, uvy, 0.0f, 1.0f, d);
    d_mul44(P.camera2World, d[0], d[1], d[2], 0.0f, dw);
    origin = rt_mk3(o[0], o[1], o[2]);
    direction = rt_normalize3(rt_mk3(dw[0], dw[1], dw[2]));
}

RTD void d_dof_camera_ray(const RtParams &P, rt_rng &rng, float uvx, float uvy, f3 &origin, f3 &direction)
{
    float d[4];
    d_mul44(P.cameraInverseProjection, uvx, uvy, 0.0f, 1.0f, d);
    f3 dir = rt_mk3(d[0], d[1], d[2]);
    const float ux = rt_rand(&rng), uy = rt_rand(&rng);
    // ConcentricSampleDisk (Header.cginc:126-149)
    const float u1x = ux * 2.0f - 1, u1y = uy * 2.0f - 1;
    float offx = 0, offy = 0;
    if (!(u1x == 0 && u1y == 0)) {
        float theta, r;
        if (rt_abs(u1x) > rt_abs(u1y)) { r = u1x; theta = u1y / u1x * RT_PI / 4; }
        else { r = u1y; theta = RT_PI / 2 - u1x / u1y * RT_PI / 4; }
        offx = r * rt_cos(theta);
        offy = r * rt_sin(theta);
    }
    offx *= P.lensRadius;
    offy *= P.lensRadius;
    const float ft = rt_abs(P.focalLength / dir.z);
    const f3 focusPoint = rt_add3(rt_splat3(0.0f), rt_scale3(dir, ft));
    const f3 org = rt_mk3(offx, offy, 0);
    dir = rt_normalize3(rt_sub3(focusPoint, org));
    float ow[4], dw[4];
    d_mul44(P.camera2World, org.x, org.y, org.z, 1, ow);
    d_mul44(P.camera2World, dir.x, dir.y, dir.z, 0.0f, dw);
    origin = rt_mk3(ow[0], ow[1], ow[2]);
    direction = rt_normalize3(rt_mk3(dw[0], dw[1], dw[2]));
}

#endif
