// rt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's GPU hot path (LBVH build, BVH traversal,
// Monte Carlo shading, accumulation), written function-by-function after the
// HLSL, serial and literal on purpose.  It exists only so that tests/, the
// smoke check and bench.py's cpu_baseline leg can compare the HIP library's
// results against it; nothing in raytracing_amd/ may link, import or call it.
//
// Parity pinning: the reference ships no tests, golden vectors or fixtures for
// this path (no test directory, no [Test] attribute anywhere under Assets/) and
// its shaders cannot be compiled or run in this environment (Unity + DXC +
// a D3D/Vulkan device are required).  PARITY UNPINNED at the level of exact
// values: this restatement is checked against hand-derived known answers and
// brute-force properties (tests/test_oracle_cpu.py).  The only outputs of the
// reference that exist are three images its author published
// (Assets/Outputs/CornellBox.png, MIS.png, LightSampling.png) for scenes that
// are in the checkout; renders of those scenes agree with them to a few 8-bit
// counts in 16 x 16 block means (tests/test_ingest_cpu.py,
// tests/test_gpu_scale.py) -- an anchor on geometry, camera, materials, light
// sampling and MIS, not on individual samples.
//
// Numeric rules the reference leaves to the GPU vendor are fixed in
// include/rt_math.h (shared with the kernels so both sides agree bit for bit).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fno-fast-math -fopenmp).

#include "../include/rtcore.h"
#include "../include/rt_math.h"
#include "../include/rt_qh.h"
#include "../include/rt_finekey.h"
#include "../include/rt_rotate.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef rt_f3 f3;

namespace {

struct Ray {            // Header.cginc:32-38
    f3 origin, direction, energy;
};

struct RayHit {         // Header.cginc:40-58
    f3 position;
    float distance;
    f3 normal;
    f3 albedo;
    float metallic, smoothness, transparent;
    f3 emissionColor;
    uint32_t materialType;
    int castShadow, receiveShadow;
    float ior;
    f3 clearCoatColor;
    uint32_t triangle;  // not in the reference struct: original triangle id, for checks
    float u, v;
};

struct TraceCounters {
    uint64_t rays, internalVisits, leafVisits, triTests, hits;
    uint32_t *nodeVisits = nullptr;   // optional per-node histogram of passed internal nodes (analysis only)
};

struct Scene {
    std::vector<RtMeshVertex> verts;
    std::vector<uint32_t> idx, matIdx;
    std::vector<int32_t> shadow;       // 2 per triangle
    uint32_t N = 0;
    std::vector<RtMaterialData> mats;
    std::vector<RtSphereLight> sphereLights;
    std::vector<RtAreaLight> areaLights;
    std::vector<RtDiscLight> discLights;
    std::vector<float> cube;           // 6 * size * size * 4
    uint32_t cubeSize = 0;

    // brute-force primitives (MeshCollector.cs:170-200)
    std::vector<RtPlane> planeBuffer;
    std::vector<RtSphere> sphereBuffer;
    std::vector<RtCube> cubeBuffer;
    std::vector<RtCustomMesh> customMeshBuffer;
    std::vector<f3> vertexBuffer, normalBuffer;
    std::vector<int32_t> indexBuffer;

    // build products (wire structs)
    std::vector<RtAABB> triAABB;
    std::vector<RtTriangle> tris;
    std::vector<uint32_t> mortonUnsorted, mortonSortedRaw, morton, sortedIdx;
    std::vector<RtInternalNode> inodes;
    std::vector<RtLeafNode> leaves;
    std::vector<RtAABB> bvhData;
    bool built = false;

    // The library's own traversal structures, rebuilt here so that its work counters can be checked:
    // a second tree over the same triangle boxes, made by the same steps from the keys of
    // include/rt_finekey.h (only N, triAABB and the build products of `walk` are filled), its child
    // boxes on the half-float grid of include/rt_qh.h, and every triangle's place in the reference's
    // visiting order (decides between equally distant triangles when the walk order differs).
    std::shared_ptr<Scene> walk;
    RtQGrid grid;
    std::vector<uint32_t> qboxes;      // 6 words per internal node of `walk`: left box x, y, z, right box x, y, z (low | high << 16)
    std::vector<uint32_t> visitRank;   // per original triangle
    bool gridOk = true;
};

// Per-thread "globals" of the shader (Header.cginc:61-64) + uniforms.
struct Thread {
    const Scene *sc;
    const RtParams *P;
    rt_rng rng;
    TraceCounters *cnt;
};

inline f3 ld3(const float *p) { return rt_mk3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------ a1
// MeshData.compute:41-48
uint32_t ExpandBits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// MeshData.compute:51-60
uint32_t MortonCode(f3 v)
{
    v.x = rt_min(rt_max(v.x * 1024.0f, 0.0f), 1023.0f);
    v.y = rt_min(rt_max(v.y * 1024.0f, 0.0f), 1023.0f);
    v.z = rt_min(rt_max(v.z * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = ExpandBits(rt_ftou(v.x));
    uint32_t yy = ExpandBits(rt_ftou(v.y));
    uint32_t zz = ExpandBits(rt_ftou(v.z));
    return xx * 4 + yy * 2 + zz;
}

// MeshData.compute:62-131 (one iteration = one thread)
void Calculate(Scene &s)
{
    const uint32_t N = s.N;
    s.triAABB.assign(N, RtAABB{});
    s.tris.assign(N, RtTriangle{});
    s.mortonUnsorted.assign(N, 0);
    s.sortedIdx.assign(N, 0);
    // MeshBufferContainer.cs:9-14,73-77: fixed +-125 box, not the mesh bounds
    const f3 encMin = rt_mk3(-125.0f, -125.0f, -125.0f);
    const f3 encMax = rt_mk3(125.0f, 125.0f, 125.0f);
    for (uint32_t id = 0; id < N; ++id) {
        const uint32_t i0 = s.idx[id * 3 + 0], i1 = s.idx[id * 3 + 1], i2 = s.idx[id * 3 + 2];
        const RtMeshVertex &a = s.verts[i0], &b = s.verts[i1], &c = s.verts[i2];
        const f3 p0 = ld3(a.position), p1 = ld3(b.position), p2 = ld3(c.position);
        RtAABB box;
        memset(&box, 0, sizeof box);
        box.min[0] = rt_min(rt_min(p0.x, p1.x), p2.x) - 0.001f;
        box.min[1] = rt_min(rt_min(p0.y, p1.y), p2.y) - 0.001f;
        box.min[2] = rt_min(rt_min(p0.z, p1.z), p2.z) - 0.001f;
        box.max[0] = rt_max(rt_max(p0.x, p1.x), p2.x) + 0.001f;
        box.max[1] = rt_max(rt_max(p0.y, p1.y), p2.y) + 0.001f;
        box.max[2] = rt_max(rt_max(p0.z, p1.z), p2.z) + 0.001f;
        s.triAABB[id] = box;

        const f3 center = rt_scale3(rt_add3(ld3(box.min), ld3(box.max)), 0.5f);
        const f3 normalized = rt_div3(rt_sub3(center, encMin), rt_sub3(encMax, encMin));
        s.mortonUnsorted[id] = MortonCode(normalized);
        s.sortedIdx[id] = id;

        RtTriangle t;
        memset(&t, 0, sizeof t);
        memcpy(t.point0, a.position, 12); memcpy(t.point1, b.position, 12); memcpy(t.point2, c.position, 12);
        memcpy(t.normal0, a.normal, 12);  memcpy(t.normal1, b.normal, 12);  memcpy(t.normal2, c.normal, 12);
        memcpy(t.tangent0, a.tangent, 12); memcpy(t.tangent1, b.tangent, 12); memcpy(t.tangent2, c.tangent, 12);
        memcpy(t.uv0, a.uv, 8); memcpy(t.uv1, b.uv, 8); memcpy(t.uv2, c.uv, 8);
        t.materialIndex = (int32_t)s.matIdx[id];
        t.castShadow = s.shadow[id * 2 + 0];
        t.receiveShadow = s.shadow[id * 2 + 0];   // MeshData.compute:127-128: [0] twice
        s.tris[id] = t;
    }
}

// ------------------------------------------------------------------ a2
// LocalRadixSort.compute / Scan.compute / GlobalRadixSort.compute driven by
// ComputeBufferSorter.cs:105-131: four passes of an 8-bit-digit stable
// counting sort of (key,value) pairs, least significant digit first.
// (The reference's in-block ranking assumes 32-lane waves; what it computes on
// such hardware is exactly this stable LSD sort.)
void RadixSort(std::vector<uint32_t> &keys, std::vector<uint32_t> &vals)
{
    const size_t n = keys.size();
    std::vector<uint32_t> k2(n), v2(n);
    for (int bitOffset = 0; bitOffset < 32; bitOffset += 8) {
        size_t count[257] = {0};
        for (size_t i = 0; i < n; ++i) count[((keys[i] >> bitOffset) & 255u) + 1]++;
        for (int d = 0; d < 256; ++d) count[d + 1] += count[d];
        for (size_t i = 0; i < n; ++i) {
            const uint32_t d = (keys[i] >> bitOffset) & 255u;
            const size_t dst = count[d]++;
            k2[dst] = keys[i];
            v2[dst] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

// ------------------------------------------------------------------ a3
// MeshBufferContainer.cs:92-107
void DistributeMortonCode(std::vector<uint32_t> &values, uint32_t trianglesLength)
{
    if (trianglesLength == 0) return;
    uint32_t newCurrentValue = 0;
    uint32_t oldCurrentValue = values[0];
    values[0] = newCurrentValue;
    for (uint32_t i = 1; i < trianglesLength; i++) {
        newCurrentValue += std::max<uint32_t>(values[i] - oldCurrentValue, 1u);
        oldCurrentValue = values[i];
        values[i] = newCurrentValue;
    }
}

// ------------------------------------------------------------------ a4
// BVH.compute:18-21
inline uint32_t clz32(uint32_t v) { return v == 0 ? 32u : (uint32_t)__builtin_clz(v); }

// BVH.compute:23-33
inline int delta(const uint32_t *codes, int x, int y, int numObjects)
{
    if (x >= 0 && x <= numObjects - 1 && y >= 0 && y <= numObjects - 1) {
        const uint32_t x_code = codes[x];
        const uint32_t y_code = codes[y];
        return (int)clz32(x_code ^ y_code);
    }
    return -1;
}

// BVH.compute:35-52.  lmax and t are uint in the shader, so products with the
// int d are evaluated modulo 2^32 and reinterpreted as int.
inline void DetermineRange(const uint32_t *codes, int numObjects, int idx, int *first, int *last)
{
    const int dd = delta(codes, idx, idx + 1, numObjects) - delta(codes, idx, idx - 1, numObjects);
    const int d = dd > 0 ? 1 : (dd < 0 ? -1 : 0);
    const int dmin = delta(codes, idx, idx - d, numObjects);
    uint32_t lmax = 2;
    while (delta(codes, idx, (int)((uint32_t)idx + lmax * (uint32_t)d), numObjects) > dmin)
        lmax = lmax * 2;
    int l = 0;
    for (uint32_t t = lmax / 2; t >= 1; t /= 2) {
        if (delta(codes, idx, (int)((uint32_t)idx + ((uint32_t)l + t) * (uint32_t)d), numObjects) > dmin)
            l += (int)t;
    }
    const int j = idx + l * d;
    *first = std::min(idx, j);
    *last = std::max(idx, j);
}

// BVH.compute:54-92
inline int FindSplit(const uint32_t *codes, int first, int last)
{
    const uint32_t firstCode = codes[first];
    const uint32_t lastCode = codes[last];
    if (firstCode == lastCode) return (first + last) >> 1;
    const int commonPrefix = (int)clz32(firstCode ^ lastCode);
    int split = first;
    int step = last - first;
    do {
        step = (step + 1) >> 1;
        const int newSplit = split + step;
        if (newSplit < last) {
            const uint32_t splitCode = codes[newSplit];
            const int splitPrefix = (int)clz32(firstCode ^ splitCode);
            if (splitPrefix > commonPrefix) split = newSplit;
        }
    } while (step > 1);
    return split;
}

// BVH.compute:94-142; buffers start as all-0xFFFFFFFF (SceneDataTypes.cs:94-119)
void TreeConstructor(Scene &s)
{
    const uint32_t N = s.N;
    RtInternalNode nullI = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    RtLeafNode nullL = {0xFFFFFFFFu, 0xFFFFFFFFu};
    s.inodes.assign(N > 0 ? N - 1 : 0, nullI);
    s.leaves.assign(N, nullL);
    if (N < 2) return;
    const uint32_t *codes = s.morton.data();
    for (uint32_t threadId = 0; threadId < N - 1; ++threadId) {
        int first, last;
        DetermineRange(codes, (int)N, (int)threadId, &first, &last);
        const int split = FindSplit(codes, first, last);
        s.inodes[threadId].index = threadId;
        if (split == first) {
            s.leaves[split].parent = threadId;
            s.leaves[split].index = (uint32_t)split;
            s.inodes[threadId].leftNode = (uint32_t)split;
            s.inodes[threadId].leftNodeType = RT_LEAF_NODE;
        } else {
            s.inodes[split].parent = threadId;
            s.inodes[threadId].leftNode = (uint32_t)split;
            s.inodes[threadId].leftNodeType = RT_INTERNAL_NODE;
        }
        if (split + 1 == last) {
            s.leaves[split + 1].parent = threadId;
            s.leaves[split + 1].index = (uint32_t)(split + 1);
            s.inodes[threadId].rightNode = (uint32_t)(split + 1);
            s.inodes[threadId].rightNodeType = RT_LEAF_NODE;
        } else {
            s.inodes[split + 1].parent = threadId;
            s.inodes[threadId].rightNode = (uint32_t)(split + 1);
            s.inodes[threadId].rightNodeType = RT_INTERNAL_NODE;
        }
    }
}

// ------------------------------------------------------------------ a5
// BVH.compute:144-162
RtAABB MergeAABB(const RtAABB &l, const RtAABB &r)
{
    RtAABB o;
    for (int k = 0; k < 3; ++k) {
        o.min[k] = rt_min(l.min[k], r.min[k]);
        o.max[k] = rt_max(l.max[k], r.max[k]);
    }
    o._dummy0 = 0;
    o._dummy1 = 0;
    return o;
}

// BVH.compute:164-211.  Threads run here one after another; the per-node flag
// (atomicsData) lets the second arriver compute the union, as in the shader.
void BVHConstructor(Scene &s)
{
    const uint32_t N = s.N;
    s.bvhData.assign(N > 0 ? N - 1 : 0, RtAABB{});   // zero-filled (MeshBufferContainer.cs:69)
    if (N < 2) return;
    std::vector<uint32_t> atomicsData(N, 0);
    for (uint32_t threadId = 0; threadId < N; ++threadId) {
        uint32_t parent = s.leaves[threadId].parent;
        while (parent != 0xFFFFFFFFu) {
            const uint32_t old = atomicsData[parent];
            if (old == 0) { atomicsData[parent] = 1; break; }
            const RtInternalNode &n = s.inodes[parent];
            const RtAABB leftAABB = n.leftNodeType == RT_INTERNAL_NODE ? s.bvhData[n.leftNode]
                                                                       : s.triAABB[s.sortedIdx[n.leftNode]];
            const RtAABB rightAABB = n.rightNodeType == RT_INTERNAL_NODE ? s.bvhData[n.rightNode]
                                                                         : s.triAABB[s.sortedIdx[n.rightNode]];
            s.bvhData[parent] = MergeAABB(leftAABB, rightAABB);
            parent = n.parent;
        }
    }
}

void RotateWalkTree(Scene &w);

// The library's traversal boxes: every child box of the tree rounded outward onto
// the half-float grid of the root box (include/rt_qh.h), 3 words per box.
// The tree the library's rays walk (no reference counterpart): the reference's own steps on finer keys.
void BuildWalkTree(Scene &s)
{
    s.walk.reset();
    if (s.N < 2) return;
    s.walk = std::make_shared<Scene>();
    Scene &w = *s.walk;
    w.N = s.N;
    w.triAABB = s.triAABB;
    const RtKeyPlan plan = rt_key_plan(s.bvhData[0].min, s.bvhData[0].max);
    std::vector<uint32_t> keys(s.N);
    w.sortedIdx.resize(s.N);
    for (uint32_t i = 0; i < s.N; ++i) {
        const RtAABB &b = s.triAABB[i];
        keys[i] = rt_fine_key(plan, b.min, b.max);
        w.sortedIdx[i] = i;
    }
    RadixSort(keys, w.sortedIdx);
    DistributeMortonCode(keys, s.N);
    w.morton = keys;
    TreeConstructor(w);
    BVHConstructor(w);
    for (int pass = 0; pass < RT_ROTATION_PASSES; ++pass) RotateWalkTree(w);
}

// Tree rotations on the walk tree (no reference counterpart; include/rt_rotate.h): one bottom-up pass, every
// node after its children.
void RotateWalkTree(Scene &w)
{
    const uint32_t N = w.N;
    if (N < 3) return;
    auto boxOf = [&](uint32_t node, uint32_t type, float out[6]) {
        const RtAABB &b = type == RT_INTERNAL_NODE ? w.bvhData[node] : w.triAABB[w.sortedIdx[w.leaves[node].index]];
        for (int k = 0; k < 3; ++k) out[k] = b.min[k], out[3 + k] = b.max[k];
    };
    auto setParent = [&](uint32_t node, uint32_t type, uint32_t parent) {
        if (type == RT_INTERNAL_NODE) w.inodes[node].parent = parent;
        else w.leaves[node].parent = parent;
    };
    auto refit = [&](uint32_t node) {
        const RtInternalNode &n = w.inodes[node];
        float a[6], b[6], m[6];
        boxOf(n.leftNode, n.leftNodeType, a);
        boxOf(n.rightNode, n.rightNodeType, b);
        rt_box_merge(a, b, m);
        RtAABB &o = w.bvhData[node];
        for (int k = 0; k < 3; ++k) o.min[k] = m[k], o.max[k] = m[3 + k];
    };
    std::vector<uint32_t> order;
    order.reserve(N - 1);
    std::vector<uint32_t> stack{0};
    while (!stack.empty()) {
        const uint32_t i = stack.back();
        stack.pop_back();
        order.push_back(i);
        const RtInternalNode &n = w.inodes[i];
        if (n.leftNodeType == RT_INTERNAL_NODE) stack.push_back(n.leftNode);
        if (n.rightNodeType == RT_INTERNAL_NODE) stack.push_back(n.rightNode);
    }
    for (size_t k = order.size(); k-- > 0;) {      // reversed pre-order: children before parents
        const uint32_t x = order[k];
        RtInternalNode &X = w.inodes[x];
        const bool lInt = X.leftNodeType == RT_INTERNAL_NODE, rInt = X.rightNodeType == RT_INTERNAL_NODE;
        float L[6], R[6], La[6] = {0}, Lb[6] = {0}, Ra[6] = {0}, Rb[6] = {0};
        boxOf(X.leftNode, X.leftNodeType, L);
        boxOf(X.rightNode, X.rightNodeType, R);
        if (lInt) {
            const RtInternalNode &c = w.inodes[X.leftNode];
            boxOf(c.leftNode, c.leftNodeType, La);
            boxOf(c.rightNode, c.rightNodeType, Lb);
        }
        if (rInt) {
            const RtInternalNode &c = w.inodes[X.rightNode];
            boxOf(c.leftNode, c.leftNodeType, Ra);
            boxOf(c.rightNode, c.rightNodeType, Rb);
        }
        const int which = rt_pick_rotation(lInt, rInt, L, R, La, Lb, Ra, Rb);
        if (which == 1 || which == 2) {
            const uint32_t r = X.rightNode;
            RtInternalNode &C = w.inodes[r];
            uint32_t &gn = which == 1 ? C.leftNode : C.rightNode, &gt = which == 1 ? C.leftNodeType : C.rightNodeType;
            std::swap(X.leftNode, gn);
            std::swap(X.leftNodeType, gt);
            setParent(X.leftNode, X.leftNodeType, x);
            setParent(gn, gt, r);
            refit(r);
        } else if (which == 5 || which == 6) {
            RtInternalNode &A = w.inodes[X.leftNode], &B = w.inodes[X.rightNode];
            uint32_t &gn = which == 5 ? B.leftNode : B.rightNode, &gt = which == 5 ? B.leftNodeType : B.rightNodeType;
            std::swap(A.leftNode, gn);
            std::swap(A.leftNodeType, gt);
            setParent(A.leftNode, A.leftNodeType, X.leftNode);
            setParent(gn, gt, X.rightNode);
            refit(X.leftNode);
            refit(X.rightNode);
        } else if (which == 3 || which == 4) {
            const uint32_t l = X.leftNode;
            RtInternalNode &C = w.inodes[l];
            uint32_t &gn = which == 3 ? C.leftNode : C.rightNode, &gt = which == 3 ? C.leftNodeType : C.rightNodeType;
            std::swap(X.rightNode, gn);
            std::swap(X.rightNodeType, gt);
            setParent(X.rightNode, X.rightNodeType, x);
            setParent(gn, gt, l);
            refit(l);
        }
    }
}

// Order in which IntersectTriangle (BVHRayTracing.cginc:78-123) meets the triangles when every box passes.
void BuildVisitRank(Scene &s)
{
    s.visitRank.assign(s.N, 0);
    if (s.N < 2) return;
    uint32_t next = 0;
    std::vector<uint32_t> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const RtInternalNode n = s.inodes[stack.back()];
        stack.pop_back();
        if (n.leftNodeType == RT_INTERNAL_NODE) stack.push_back(n.leftNode);
        else s.visitRank[s.sortedIdx[s.leaves[n.leftNode].index]] = next++;
        if (n.rightNodeType == RT_INTERNAL_NODE) stack.push_back(n.rightNode);
        else s.visitRank[s.sortedIdx[s.leaves[n.rightNode].index]] = next++;
    }
}

void BuildGridBoxes(Scene &ref)
{
    ref.qboxes.clear();
    ref.gridOk = true;
    if (ref.N < 2) return;
    ref.grid = rt_qh_grid(ref.bvhData[0].min, ref.bvhData[0].max);
    ref.qboxes.resize((size_t)(ref.N - 1) * 6);
    const Scene &s = *ref.walk;
    for (uint32_t i = 0; i + 1 < s.N; ++i) {
        const RtInternalNode &n = s.inodes[i];
        const uint32_t child[2] = {n.leftNode, n.rightNode}, type[2] = {n.leftNodeType, n.rightNodeType};
        for (int c = 0; c < 2; ++c) {
            const RtAABB &b = type[c] == RT_INTERNAL_NODE ? s.bvhData[child[c]] : s.triAABB[s.sortedIdx[s.leaves[child[c]].index]];
            if (!rt_qh_box(ref.grid, b.min, b.max, &ref.qboxes[(size_t)i * 6 + c * 3])) ref.gridOk = false;
        }
    }
}

void Build(Scene &s)
{
    Calculate(s);
    std::vector<uint32_t> keys = s.mortonUnsorted;
    RadixSort(keys, s.sortedIdx);
    s.mortonSortedRaw = keys;
    DistributeMortonCode(keys, s.N);
    s.morton = keys;
    TreeConstructor(s);
    BVHConstructor(s);
    BuildWalkTree(s);
    BuildVisitRank(s);
    BuildGridBoxes(s);
    s.built = true;
}

// ------------------------------------------------------------------ a6
// Header.cginc:173-180; fields the shader leaves unset read as zero here.
RayHit CreateRayHit()
{
    RayHit hit;
    memset(&hit, 0, sizeof hit);
    hit.distance = rt_inf();
    hit.triangle = 0xFFFFFFFFu;
    return hit;
}

// BVHRayTracing.cginc:27-39
bool RayBoxIntersection(const RtAABB &b, const Ray &r)
{
    const f3 t1 = rt_div3(rt_sub3(ld3(b.min), r.origin), r.direction);
    const f3 t2 = rt_div3(rt_sub3(ld3(b.max), r.origin), r.direction);
    const f3 tmin1 = rt_mk3(rt_min(t1.x, t2.x), rt_min(t1.y, t2.y), rt_min(t1.z, t2.z));
    const f3 tmax1 = rt_mk3(rt_max(t1.x, t2.x), rt_max(t1.y, t2.y), rt_max(t1.z, t2.z));
    const float tmin = rt_max(tmin1.x, rt_max(tmin1.y, tmin1.z));
    const float tmax = rt_min(tmax1.x, rt_min(tmax1.y, tmax1.z));
    return tmax > tmin && tmax > 0;
}

// BruteForceRayTracing.cginc:166-204
bool IntersectTriangle_MT97(const Ray &ray, f3 vert0, f3 vert1, f3 vert2, float &t, float &u, float &v)
{
    const f3 edge1 = rt_sub3(vert1, vert0);
    const f3 edge2 = rt_sub3(vert2, vert0);
    const f3 pvec = rt_cross3(ray.direction, edge2);
    const float det = rt_dot3(edge1, pvec);
    if (det < RT_EPSILON) return false;
    const float inv_det = 1.0f / det;
    const f3 tvec = rt_sub3(ray.origin, vert0);
    u = rt_dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    const f3 qvec = rt_cross3(tvec, edge1);
    v = rt_dot3(ray.direction, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = rt_dot3(edge2, qvec) * inv_det;
    return true;
}

// BVHRayTracing.cginc:41-76
// visitRank (library walk only, nullptr for the reference's own walk): among equally distant triangles the
// one the reference meets first has to win, whatever order this walk meets them in.
void CheckTriangle(const Scene &s, uint32_t triangleIndex, const Ray &ray, RayHit &hit, TraceCounters *cnt,
                   const std::vector<uint32_t> *visitRank = nullptr)
{
    if (cnt) cnt->leafVisits++;
    if (RayBoxIntersection(s.triAABB[triangleIndex], ray)) {
        if (cnt) cnt->triTests++;
        const RtTriangle &tri = s.tris[triangleIndex];
        float t = 0, u = 0, v = 0;
        if (IntersectTriangle_MT97(ray, ld3(tri.point0), ld3(tri.point1), ld3(tri.point2), t, u, v)) {
            bool closer = t > 0 && t < hit.distance;
            if (visitRank && t > 0 && t == hit.distance && hit.triangle != 0xFFFFFFFFu)
                closer = (*visitRank)[triangleIndex] < (*visitRank)[hit.triangle];
            if (closer) {
                hit.distance = t;
                hit.position = rt_add3(ray.origin, rt_scale3(ray.direction, t));
                const f3 nsum = rt_add3(rt_add3(rt_scale3(ld3(tri.normal0), (1 - u) - v),
                                                rt_scale3(ld3(tri.normal1), u)),
                                        rt_scale3(ld3(tri.normal2), v));
                hit.normal = rt_normalize3(nsum);
                RtMaterialData m;
                if ((uint32_t)tri.materialIndex < s.mats.size()) m = s.mats[tri.materialIndex];
                else memset(&m, 0, sizeof m);        // out-of-range structured-buffer read = 0
                hit.albedo = ld3(m.albedo);
                hit.metallic = m.metallic;
                hit.smoothness = m.smoothness;
                hit.transparent = m.transparent;
                hit.emissionColor = ld3(m.emissionColor);
                hit.materialType = m.materialType;
                hit.ior = m.ior;
                hit.clearCoatColor = ld3(m.clearCoatColor);
                hit.castShadow = tri.castShadow;
                hit.receiveShadow = tri.receiveShadow;
                hit.triangle = triangleIndex;
                hit.u = u;
                hit.v = v;
            }
        }
    }
}

// BVHRayTracing.cginc:78-123
void IntersectTriangle(const Scene &s, const Ray &ray, RayHit &hit, TraceCounters *cnt)
{
    if (s.N == 0) return;
    // With a single triangle there is no internal node: bvhData[0] is the
    // zero-filled box, which no ray passes, so nothing is ever hit.
    static const RtAABB zeroBox = {};
    uint32_t stack[64];
    uint32_t currentStackIndex = 0;
    stack[currentStackIndex] = 0;
    currentStackIndex = 1;
    while (currentStackIndex != 0) {
        currentStackIndex--;
        const uint32_t index = stack[currentStackIndex];
        if (cnt) cnt->internalVisits++;
        const RtAABB &box = s.N >= 2 ? s.bvhData[index] : zeroBox;
        if (!RayBoxIntersection(box, ray)) continue;
        if (cnt && cnt->nodeVisits) __atomic_fetch_add(&cnt->nodeVisits[index], 1u, __ATOMIC_RELAXED);
        const RtInternalNode &n = s.inodes[index];
        if (n.leftNodeType == RT_INTERNAL_NODE) {
            stack[currentStackIndex] = n.leftNode;
            currentStackIndex++;
        } else {
            const uint32_t triangleIndex = s.sortedIdx[s.leaves[n.leftNode].index];
            CheckTriangle(s, triangleIndex, ray, hit, cnt);
        }
        if (n.rightNodeType == RT_INTERNAL_NODE) {
            stack[currentStackIndex] = n.rightNode;
            currentStackIndex++;
        } else {
            const uint32_t triangleIndex = s.sortedIdx[s.leaves[n.rightNode].index];
            CheckTriangle(s, triangleIndex, ray, hit, cnt);
        }
        if (currentStackIndex > 64) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
    }
}

// ------------------------------------------------------------------ grid walk (no reference counterpart)
// Restates the library's one-ray-per-lane walker on the CPU so that its work
// counters can be compared exactly and its claim "same hit as the reference's
// walk" can be checked on the CPU as well: child boxes rounded outward onto the
// half-float grid of the root box and tested with rt_slab_maybe; the triangle's own
// box exact, with the reference's test; rays with a zero / non-finite component
// use the exact boxes and the reference's test everywhere.
const RtAABB &ChildBox(const Scene &s, uint32_t node, uint32_t type)
{
    return type == RT_INTERNAL_NODE ? s.bvhData[node] : s.triAABB[s.sortedIdx[s.leaves[node].index]];
}

// Analysis aid (orc_set_prune_scale): while finite, the grid walk goes front to back and does not enter a box whose
// entry distance exceeds (closest hit so far) x this -- the library's RT_OPT_PRUNE_BY_DISTANCE rule, for finding on
// the CPU the rays whose hit that rule changes.  +inf (the default): the walk above, every box the ray passes.
static float g_pruneScale = rt_inf();

void IntersectTriangleGrid(const Scene &ref, const Ray &ray, RayHit &hit, TraceCounters *cnt)
{
    if (ref.N == 0) return;
    if (cnt) cnt->internalVisits++;
    if (ref.N < 2 || !RayBoxIntersection(ref.bvhData[0], ray)) return;
    // a ray with a zero / non-finite direction component (or slab constants out of range) walks the reference's
    // tree with the reference's test
    const f3 r = rt_mk3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    const RtRaySlab slab = rt_ray_slab(ref.grid, ray.origin, r);
    const bool odd = rt_ray_needs_exact_walk(ray.origin, ray.direction) || !rt_slab_finite(slab) || !rt_slab_rejects_empty_box(slab);
    const Scene &s = odd ? ref : *ref.walk;
    uint32_t stack[64];
    uint32_t sp = 0;
    stack[sp++] = 0;
    while (sp != 0) {
        const uint32_t index = stack[--sp];
        const RtInternalNode &n = s.inodes[index];
        if (cnt) cnt->internalVisits += 2;
        bool pass[2];
        float tnear[2] = {0.0f, 0.0f};
        const uint32_t child[2] = {n.leftNode, n.rightNode}, type[2] = {n.leftNodeType, n.rightNodeType};
        for (int c = 0; c < 2; ++c) {
            if (odd) {
                pass[c] = RayBoxIntersection(ChildBox(s, child[c], type[c]), ray);
            } else {
                const uint32_t *w = &ref.qboxes[(size_t)index * 6 + c * 3];
                pass[c] = rt_slab_maybe_near(slab, w[0], w[1], w[2], &tnear[c]);
                if (g_pruneScale < rt_inf() && !(tnear[c] <= hit.distance * g_pruneScale)) pass[c] = false;
            }
        }
        // left leaf, right leaf, then the right subtree before the left one
        for (int c = 0; c < 2; ++c)
            if (type[c] != RT_INTERNAL_NODE && pass[c]) CheckTriangle(ref, s.sortedIdx[s.leaves[child[c]].index], ray, hit, cnt, &ref.visitRank);
        const bool leftFirst = g_pruneScale < rt_inf() && !odd && tnear[0] < tnear[1];   // (the one pushed last is walked first)
        for (int k = 0; k < 2; ++k) {
            const int c = leftFirst ? 1 - k : k;
            if (type[c] == RT_INTERNAL_NODE && pass[c]) stack[sp++] = child[c];
        }
        if (sp > 62) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
    }
}

// The same walk two levels of the tree at a time, as the library's default walker takes it (no reference
// counterpart): a step looks at a node's GRANDCHILDREN -- of a child that is a leaf, the child itself -- through the
// boxes their parents hold, so a child's own box is never tested.  Inner boxes only filter, so the hit is the same;
// what differs is the work: every place of a step that holds a child counts as one box tested.
void IntersectTriangleWide(const Scene &ref, const Ray &ray, RayHit &hit, TraceCounters *cnt)
{
    const f3 r = rt_mk3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    const RtRaySlab slab = rt_ray_slab(ref.grid, ray.origin, r);
    if (ref.N < 2 || rt_ray_needs_exact_walk(ray.origin, ray.direction) || !rt_slab_finite(slab) || !rt_slab_rejects_empty_box(slab)) {
        IntersectTriangleGrid(ref, ray, hit, cnt);       // the exact walk of the reference's tree, as there
        return;
    }
    if (cnt) cnt->internalVisits++;
    if (!RayBoxIntersection(ref.bvhData[0], ray)) return;
    const Scene &s = *ref.walk;
    std::vector<uint32_t> stack{0};
    while (!stack.empty()) {
        const uint32_t index = stack.back();
        stack.pop_back();
        const RtInternalNode &n = s.inodes[index];
        const uint32_t child[2] = {n.leftNode, n.rightNode}, type[2] = {n.leftNodeType, n.rightNodeType};
        for (int c = 0; c < 2; ++c) {
            // the places this child fills: itself with the box its parent holds, or its two children with the boxes it holds
            uint32_t place[2], placeType[2];
            const uint32_t *box[2];
            int nPlaces = 0;
            if (type[c] != RT_INTERNAL_NODE) {
                place[0] = child[c]; placeType[0] = type[c]; box[0] = &ref.qboxes[(size_t)index * 6 + c * 3];
                nPlaces = 1;
            } else {
                const RtInternalNode &g = s.inodes[child[c]];
                place[0] = g.leftNode; placeType[0] = g.leftNodeType; box[0] = &ref.qboxes[(size_t)child[c] * 6];
                place[1] = g.rightNode; placeType[1] = g.rightNodeType; box[1] = &ref.qboxes[(size_t)child[c] * 6 + 3];
                nPlaces = 2;
            }
            for (int k = 0; k < nPlaces; ++k) {
                if (cnt) cnt->internalVisits++;
                if (!rt_slab_maybe(slab, box[k][0], box[k][1], box[k][2])) continue;
                if (placeType[k] == RT_INTERNAL_NODE) stack.push_back(place[k]);
                else CheckTriangle(ref, s.sortedIdx[s.leaves[place[k]].index], ray, hit, cnt, &ref.visitRank);
            }
        }
    }
}

// BVHRayTracing.cginc:125-133
RayHit BVHRayTrace(const Thread &th, const Ray &ray)
{
    RayHit hit = CreateRayHit();
    if (th.cnt) th.cnt->rays++;
    IntersectTriangle(*th.sc, ray, hit, th.cnt);
    if (th.cnt && hit.distance < rt_inf()) th.cnt->hits++;
    return hit;
}

// 4x4 (Unity memory order) times float4; rows summed left to right.
inline void mul44(const float *m, float x, float y, float z, float w, float out[4])
{
    for (int r = 0; r < 4; ++r)
        out[r] = ((m[0 * 4 + r] * x + m[1 * 4 + r] * y) + m[2 * 4 + r] * z) + m[3 * 4 + r] * w;
}

// ------------------------------------------------------------------ BruteForceRayTracing.cginc
// The RayHit fields only IntersectPlane writes (materialType, ior, clearCoatColor, shadow flags)
// keep whatever an earlier primitive left there, as in the shader; `primitive` (not in the
// reference) remembers who wrote the hit last: kind << 28 | index.
inline f3 mulPoint(const float *m, f3 v, float w)
{
    float out[4];
    mul44(m, v.x, v.y, v.z, w, out);
    return rt_mk3(out[0], out[1], out[2]);
}

// :68-99
void IntersectPlane(const Ray &ray, const RtPlane &plane, uint32_t index, RayHit &hit)
{
    const f3 normal = ld3(plane.normal), position = ld3(plane.position), size = ld3(plane.size);
    const float t = rt_dot3(normal, rt_sub3(position, ray.origin)) / rt_dot3(normal, ray.direction);
    if (t > 0 && t < hit.distance) {
        const f3 mn = rt_sub3(position, rt_divs3(size, 2.0f));
        const f3 mx = rt_add3(position, rt_divs3(size, 2.0f));
        const f3 p = rt_add3(ray.origin, rt_scale3(ray.direction, t));
        if (p.x > mn.x && p.x < mx.x && p.y > mn.y && p.y < mx.y && p.z > mn.z && p.z < mx.z) {
            hit.distance = t;
            hit.position = p;
            hit.normal = normal;
            hit.albedo = ld3(plane.albedo);
            hit.metallic = plane.metallic;
            hit.smoothness = plane.smoothness;
            hit.transparent = plane.transparent;
            hit.emissionColor = ld3(plane.emissionColor);
            hit.materialType = 0;
            hit.ior = 1.0f;
            hit.clearCoatColor = rt_splat3(0.0f);
            hit.castShadow = 0;
            hit.receiveShadow = 0;
            hit.triangle = (RT_PRIM_PLANE << 28) | index;
        }
    }
}

// :101-130
void IntersectSphere(const Ray &ray, const RtSphere &sphere, uint32_t index, RayHit &hit)
{
    const f3 center = ld3(sphere.center);
    const float distance = rt_length3(rt_sub3(center, ray.origin));
    if (distance < sphere.radius) return;
    const f3 dir = rt_sub3(center, ray.origin);
    const float l = rt_dot3(dir, rt_normalize3(ray.direction));
    if (l < 0) return;
    const float m = rt_sqrt(distance * distance - l * l);
    if (m > sphere.radius) return;
    const float q = rt_sqrt(sphere.radius * sphere.radius - m * m);
    hit.distance = l - q;
    hit.position = rt_add3(ray.origin, rt_scale3(ray.direction, l - q));
    hit.normal = rt_normalize3(rt_sub3(hit.position, center));
    hit.albedo = ld3(sphere.albedo);
    hit.metallic = sphere.metallic;
    hit.smoothness = sphere.smoothness;
    hit.transparent = sphere.transparent;
    hit.emissionColor = ld3(sphere.emissionColor);
    hit.triangle = (RT_PRIM_SPHERE << 28) | index;
}

// :132-163
void IntersectCube(const Ray &ray, const RtCube &cube, uint32_t index, RayHit &hit)
{
    const f3 cmin = ld3(cube.min), cmax = ld3(cube.max);
    const f3 invDir = rt_div3(rt_splat3(1.0f), ray.direction);
    const f3 tMin = rt_mul3(rt_sub3(cmin, ray.origin), invDir);
    const f3 tMax = rt_mul3(rt_sub3(cmax, ray.origin), invDir);
    const f3 t1 = rt_mk3(rt_min(tMin.x, tMax.x), rt_min(tMin.y, tMax.y), rt_min(tMin.z, tMax.z));
    const f3 t2 = rt_mk3(rt_max(tMin.x, tMax.x), rt_max(tMin.y, tMax.y), rt_max(tMin.z, tMax.z));
    const float tNear = rt_max(rt_max(t1.x, t1.y), t1.z);
    const float tFar = rt_min(rt_min(t2.x, t2.y), t2.z);
    if (tNear > tFar) return;
    f3 normal = rt_mk3(1, 0, 0);
    if (tNear == t1.y) normal = rt_mk3(0, 1, 0);
    else if (tNear == t1.z) normal = rt_mk3(0, 0, 1);
    const f3 side = rt_sub3(ray.origin, rt_scale3(rt_add3(cmin, cmax), 0.5f));
    normal = rt_mul3(normal, rt_mk3(rt_sign(side.x), rt_sign(side.y), rt_sign(side.z)));
    hit.distance = tNear;
    hit.position = rt_add3(ray.origin, rt_scale3(ray.direction, tNear));
    hit.normal = normal;
    hit.albedo = ld3(cube.albedo);
    hit.metallic = cube.metallic;
    hit.smoothness = cube.smoothness;
    hit.transparent = cube.transparent;
    hit.emissionColor = ld3(cube.emissionColor);
    hit.triangle = (RT_PRIM_CUBE << 28) | index;
}

// :206-243; `firstTriangle` numbers the triangles over all meshes in list order
void IntersectCustomMesh(const Scene &s, const Ray &ray, const RtCustomMesh &mesh, uint32_t firstTriangle, RayHit &hit)
{
    const uint32_t offset = (uint32_t)mesh.indicesOffset;
    const uint32_t count = offset + (uint32_t)mesh.indicesCount;
    uint32_t tri = firstTriangle;
    for (uint32_t i = offset; i < count; i += 3, ++tri) {
        const f3 v0 = mulPoint(mesh.localToWorldMatrix, s.vertexBuffer[s.indexBuffer[i]], 1);
        const f3 v1 = mulPoint(mesh.localToWorldMatrix, s.vertexBuffer[s.indexBuffer[i + 1]], 1);
        const f3 v2 = mulPoint(mesh.localToWorldMatrix, s.vertexBuffer[s.indexBuffer[i + 2]], 1);
        const f3 n0 = mulPoint(mesh.localToWorldMatrix, s.normalBuffer[s.indexBuffer[i]], 0);
        const f3 n1 = mulPoint(mesh.localToWorldMatrix, s.normalBuffer[s.indexBuffer[i + 1]], 0);
        const f3 n2 = mulPoint(mesh.localToWorldMatrix, s.normalBuffer[s.indexBuffer[i + 2]], 0);
        float t = 0, u = 0, v = 0;
        if (IntersectTriangle_MT97(ray, v0, v1, v2, t, u, v)) {
            if (t > 0 && t < hit.distance) {
                hit.distance = t;
                hit.position = rt_add3(ray.origin, rt_scale3(ray.direction, t));
                hit.normal = rt_normalize3(rt_add3(rt_add3(rt_scale3(n0, (1 - u) - v), rt_scale3(n1, u)), rt_scale3(n2, v)));
                hit.albedo = ld3(mesh.albedo);
                hit.metallic = mesh.metallic;
                hit.smoothness = mesh.smoothness;
                hit.transparent = mesh.transparent;
                hit.emissionColor = ld3(mesh.emissionColor);
                hit.triangle = (RT_PRIM_TRIANGLE << 28) | tri;
                hit.u = u;
                hit.v = v;
            }
        }
    }
}

// :246-279
RayHit BruteForceRayTrace(const Thread &th, const Ray &ray)
{
    const Scene &s = *th.sc;
    RayHit hit = CreateRayHit();
    if (th.cnt) th.cnt->rays++;
    for (size_t i = 0; i < s.planeBuffer.size(); ++i) IntersectPlane(ray, s.planeBuffer[i], (uint32_t)i, hit);
    for (size_t i = 0; i < s.sphereBuffer.size(); ++i) IntersectSphere(ray, s.sphereBuffer[i], (uint32_t)i, hit);
    for (size_t i = 0; i < s.cubeBuffer.size(); ++i) IntersectCube(ray, s.cubeBuffer[i], (uint32_t)i, hit);
    uint32_t firstTriangle = 0;
    for (size_t i = 0; i < s.customMeshBuffer.size(); ++i) {
        IntersectCustomMesh(s, ray, s.customMeshBuffer[i], firstTriangle, hit);
        firstTriangle += (uint32_t)s.customMeshBuffer[i].indicesCount / 3;
    }
    if (th.cnt && hit.distance < rt_inf()) th.cnt->hits++;
    return hit;
}

// ------------------------------------------------------------------ a7 / a8
inline float Rand(Thread &th) { return rt_rand(&th.rng); }     // Header.cginc:86-92


// Header.cginc:99-108
Ray CreateRay(f3 origin, f3 direction)
{
    Ray ray;
    ray.origin = origin;
    ray.direction = direction;
    ray.energy = rt_mk3(1.0f, 1.0f, 1.0f);
    return ray;
}

// Header.cginc:110-122
Ray CreateCameraRay(const Thread &th, float uvx, float uvy)
{
    float o[4], d[4], dw[4];
    mul44(th.P->camera2World, 0, 0, 0, 1, o);
    mul44(th.P->cameraInverseProjection, uvx, uvy, 0.0f, 1.0f, d);
    mul44(th.P->camera2World, d[0], d[1], d[2], 0.0f, dw);
    return CreateRay(rt_mk3(o[0], o[1], o[2]), rt_normalize3(rt_mk3(dw[0], dw[1], dw[2])));
}

// Header.cginc:126-149
rt_f2 ConcentricSampleDisk(rt_f2 u)
{
    rt_f2 u1 = {u.x * 2.0f - 1, u.y * 2.0f - 1};
    rt_f2 zero = {0, 0};
    if (u1.x == 0 && u1.y == 0) return zero;
    float theta, r;
    if (rt_abs(u1.x) > rt_abs(u1.y)) {
        r = u1.x;
        theta = u1.y / u1.x * RT_PI / 4;
    } else {
        r = u1.y;
        theta = RT_PI / 2 - u1.x / u1.y * RT_PI / 4;
    }
    rt_f2 o = {r * rt_cos(theta), r * rt_sin(theta)};
    return o;
}

// Header.cginc:151-170
Ray CreateDoFCameraRay(Thread &th, float uvx, float uvy)
{
    float d[4];
    mul44(th.P->cameraInverseProjection, uvx, uvy, 0.0f, 1.0f, d);
    f3 direction = rt_mk3(d[0], d[1], d[2]);
    rt_f2 rr;
    rr.x = Rand(th);
    rr.y = Rand(th);
    rt_f2 offset = ConcentricSampleDisk(rr);
    offset.x *= th.P->lensRadius;
    offset.y *= th.P->lensRadius;
    const float ft = rt_abs(th.P->focalLength / direction.z);
    const f3 focusPoint = rt_add3(rt_splat3(0.0f), rt_scale3(direction, ft));
    f3 origin = rt_mk3(offset.x, offset.y, 0);
    direction = rt_normalize3(rt_sub3(focusPoint, origin));
    float ow[4], dw[4];
    mul44(th.P->camera2World, origin.x, origin.y, origin.z, 1, ow);
    mul44(th.P->camera2World, direction.x, direction.y, direction.z, 0.0f, dw);
    return CreateRay(rt_mk3(ow[0], ow[1], ow[2]), rt_normalize3(rt_mk3(dw[0], dw[1], dw[2])));
}

// ------------------------------------------------------------------ a9: BRDF.cginc
const float DIELECTRIC_SPEC = 0.04f;                 // BRDF.cginc:4
const float ONE_MINUS_DIELECTRIC_SPEC = 1.0f - 0.04f;

inline float Pow5(float x) { return x * x * x * x * x; }                               // :6-9
inline float OneMinusReflectivityFromMetallic(float metallic)                          // :11-20
{
    const float oneMinusDielectricSpec = ONE_MINUS_DIELECTRIC_SPEC;
    return oneMinusDielectricSpec - metallic * oneMinusDielectricSpec;
}
inline float SmoothnessToPerceptualRoughness(float smoothness) { return 1 - smoothness; } // :22-25
inline float PerceptualRoughnessToRoughness(float pr) { return pr * pr; }                 // :27-30

inline f3 DiffuseAndSpecularFromMetallic(f3 albedo, float metallic, f3 &specColor, float &oneMinusReflectivity) // :32-37
{
    specColor = rt_lerp3(rt_splat3(DIELECTRIC_SPEC), albedo, metallic);
    oneMinusReflectivity = OneMinusReflectivityFromMetallic(metallic);
    return rt_scale3(albedo, oneMinusReflectivity);
}

inline float DisneyDiffuse(float NdotV, float NdotL, float LdotH, float perceptualRoughness) // :39-47
{
    const float fd90 = 0.5f + 2 * LdotH * LdotH * perceptualRoughness;
    const float lightScatter = (1 + (fd90 - 1) * Pow5(1 - NdotL));
    const float viewScatter = (1 + (fd90 - 1) * Pow5(1 - NdotV));
    return lightScatter * viewScatter;
}

inline f3 DiffuseBRDF(f3 albedo, f3 normal, f3 viewDir, f3 halfDir, f3 lightDir, float perceptualRoughness, float &pdf) // :49-62
{
    const float nv = rt_saturate(rt_dot3(normal, viewDir));
    const float nl = rt_saturate(rt_dot3(normal, lightDir));
    const float lh = rt_saturate(rt_dot3(lightDir, halfDir));
    const float diffuseTerm = DisneyDiffuse(nv, nl, lh, perceptualRoughness) * nl;
    const f3 brdf = rt_divs3(rt_scale3(albedo, diffuseTerm), RT_PI);
    pdf = nl / RT_PI;
    return brdf;
}

inline float GGXTerm(float NdotH, float roughness)                                     // :65-71
{
    const float a2 = roughness * roughness;
    const float d = (NdotH * a2 - NdotH) * NdotH + 1.0f;
    return a2 / (RT_PI * (d * d + 1e-7f));
}

inline f3 FresnelTerm(f3 F0, float cosA)                                               // :73-77
{
    const float t = Pow5(1 - cosA);
    return rt_add3(F0, rt_scale3(rt_sub3(rt_splat3(1.0f), F0), t));
}

inline float GeometrySchlickGGX(float NdotV, float roughness)                          // :90-99
{
    const float r = (roughness + 1.0f);
    const float k = (r * r) / 8.0f;
    const float nom = NdotV;
    const float denom = NdotV * (1.0f - k) + k;
    return nom / denom;
}

inline float GeometrySmith(f3 N, f3 V, f3 L, float roughness)                          // :106-114
{
    const float NdotV = rt_saturate(rt_dot3(N, V));
    const float NdotL = rt_saturate(rt_dot3(N, L));
    const float ggx2 = GeometrySchlickGGX(NdotV, roughness);
    const float ggx1 = GeometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}

inline f3 SpecularBRDF(f3 specColor, f3 normal, f3 viewDir, f3 halfDir, f3 lightDir, float roughness, f3 &F, float &pdf) // :116-139
{
    const float nv = rt_abs(rt_dot3(normal, viewDir));
    const float nl = rt_saturate(rt_dot3(normal, lightDir));
    const float nh = rt_saturate(rt_dot3(normal, halfDir));
    const float hv = rt_saturate(rt_dot3(halfDir, viewDir));
    const float D = GGXTerm(nh, roughness);
    F = FresnelTerm(specColor, hv);
    const float G = GeometrySmith(normal, viewDir, lightDir, roughness);
    const f3 nominator = rt_scale3(F, D * G);
    const float denominator = 4.0f * nv * nl + 0.001f;
    const f3 brdf = rt_divs3(nominator, denominator);
    pdf = D * nh / (4.0f * hv);
    return brdf;
}

// BRDF.cginc:141-181
void BRDF(const Thread &th, f3 viewDir, f3 halfDir, f3 lightDir, f3 albedo, f3 normal, float metallic,
          float perceptualRoughness, float roughness, float diffuseRatio, float specularRoatio, f3 &func, float &pdf)
{
    float oneMinusReflectivity;
    f3 specColor;
    const f3 diffuse = DiffuseAndSpecularFromMetallic(albedo, metallic, specColor, oneMinusReflectivity);
    float diffusePdf;
    const f3 diffuseBRDF = DiffuseBRDF(diffuse, normal, viewDir, halfDir, lightDir, perceptualRoughness, diffusePdf);
    f3 F;
    float specularPdf;
    const f3 specularBRDF = SpecularBRDF(specColor, normal, viewDir, halfDir, lightDir, roughness, F, specularPdf);
    const float *lcol = th.P->lightColor;
    const f3 lc = rt_scale3(rt_mk3(lcol[0], lcol[1], lcol[2]), lcol[3]);
    const f3 kS = F;
    f3 kD = rt_sub3(rt_splat3(1.0f), kS);
    kD = rt_scale3(kD, 1.0f - metallic);
    const f3 a = rt_divs3(rt_mul3(rt_mul3(diffuseBRDF, kD), lc), 2.0f);
    const f3 b = rt_mul3(specularBRDF, lc);
    const f3 totalBRDF = rt_scale3(rt_add3(a, b), rt_saturate(rt_dot3(normal, lightDir)));
    const float totalPdf = diffusePdf * diffuseRatio + specularPdf * specularRoatio;
    func = totalBRDF;
    pdf = totalPdf;
}

// BRDF.cginc:184-204
float FresnelReflectAmount(float n1, float n2, f3 normal, f3 incident, float f0, float f90)
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

// ------------------------------------------------------------------ a9: ImportanceSampling.cginc
// :42-52; rows = tangent, binormal, normal
void GetTangentSpace(f3 normal, f3 rows[3])
{
    f3 helper = rt_mk3(1, 0, 0);
    if (rt_abs(normal.x) > 0.99f) helper = rt_mk3(0, 0, 1);
    const f3 tangent = rt_normalize3(rt_cross3(normal, helper));
    const f3 binormal = rt_normalize3(rt_cross3(normal, tangent));
    rows[0] = tangent;
    rows[1] = binormal;
    rows[2] = normal;
}

// mul(float3 v, float3x3 M) = v.x*row0 + v.y*row1 + v.z*row2
inline f3 mulVecMat(f3 v, const f3 rows[3])
{
    return rt_mk3((v.x * rows[0].x + v.y * rows[1].x) + v.z * rows[2].x,
                  (v.x * rows[0].y + v.y * rows[1].y) + v.z * rows[2].y,
                  (v.x * rows[0].z + v.y * rows[1].z) + v.z * rows[2].z);
}

// :55-60
f3 Tangent2World(float theta, float phi, f3 direction)
{
    float st, ct, sp, cp;
    rt_sincos(theta, &st, &ct);
    rt_sincos(phi, &sp, &cp);
    const f3 localSpaceDir = rt_mk3(cp * st, sp * st, ct);
    f3 rows[3];
    GetTangentSpace(direction, rows);
    return rt_normalize3(mulVecMat(localSpaceDir, rows));
}

const float TWO_PI = 2.0f * RT_PI;

// :62-83
f3 UniformSampling(Thread &th, const RayHit &hit, Ray &ray)
{
    const float theta = rt_acos(1 - Rand(th));
    const float phi = TWO_PI * Rand(th);
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = Tangent2World(theta, phi, hit.normal);
    const float pdf = 1.0f / TWO_PI;
    const f3 fr = rt_divs3(hit.albedo, RT_PI);
    return rt_scale3(rt_divs3(fr, pdf), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

// :86-99
f3 CosineWeightedSampling(Thread &th, const RayHit &hit, Ray &ray)
{
    const float theta = rt_acos(rt_sqrt(1 - Rand(th)));
    const float phi = TWO_PI * Rand(th);
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = Tangent2World(theta, phi, hit.normal);
    return hit.albedo;
}

// :103-118
void SphereLightImportanceSampling(Thread &th, const RayHit &hit, Ray &ray, const RtSphereLight &light, f3 &func, float &pdf)
{
    const f3 dir = rt_sub3(ld3(light.position), hit.position);
    const float maxCos = rt_sqrt(1 - rt_pow(light.radius / rt_length3(dir), 2.0f));
    const float theta = rt_acos(Rand(th) * (maxCos - 1) + 1);
    const float phi = TWO_PI * Rand(th);
    const f3 direction = rt_normalize3(dir);
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = Tangent2World(theta, phi, direction);
    pdf = 1.0f / (TWO_PI * (1 - maxCos));
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

// :147-160 (never called by the reference: _LightImportanceSampling has no disc branch)
void DiscLightImportanceSampling(Thread &th, const RayHit &hit, Ray &ray, const RtDiscLight &light, f3 &func, float &pdf)
{
    const float theta = rt_sqrt(Rand(th) * light.radius);
    const float phi = TWO_PI * Rand(th);
    const f3 direction = rt_normalize3(rt_sub3(ld3(light.position), hit.position));
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = Tangent2World(theta, phi, direction);
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
    pdf = 1 / (RT_PI * light.radius * light.radius);
}

// :122-143
void AreaLightImportanceSampling(Thread &th, const RayHit &hit, Ray &ray, const RtAreaLight &light, f3 &func, float &pdf)
{
    const float x = (Rand(th) * 2 - 1) * light.size[0] / 2;
    const float z = (Rand(th) * 2 - 1) * light.size[1] / 2;
    const f3 pointOnArea = rt_mk3(x, 0, z);
    const f3 lnormal = ld3(light.normal), lup = ld3(light.up);
    const f3 binormal = rt_normalize3(rt_cross3(lnormal, lup));
    const f3 rows[3] = {binormal, lnormal, lup};
    const f3 pointWS = rt_add3(mulVecMat(pointOnArea, rows), ld3(light.position));
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = rt_sub3(pointWS, hit.position);
    const float distanceSquard = rt_dot3(ray.direction, ray.direction);
    const float area = light.size[0] * light.size[1];
    const float lightCosine = rt_dot3(rt_normalize3(rt_neg3(ray.direction)), lnormal);
    ray.direction = rt_normalize3(ray.direction);
    pdf = distanceSquard / (lightCosine * area);
    func = rt_scale3(rt_divs3(hit.albedo, RT_PI), rt_saturate(rt_dot3(hit.normal, ray.direction)));
}

// :163-188.  With no light keyword enabled func/pdf are never written in the
// shader; here they start as func = 0, pdf = -1.
void _LightImportanceSampling(Thread &th, const RayHit &hit, Ray &ray, f3 &func, float &pdf)
{
    const Scene &s = *th.sc;
    func = rt_splat3(0.0f);
    pdf = -1.0f;
    const bool hasSphere = !s.sphereLights.empty(), hasArea = !s.areaLights.empty();
    const RtSphereLight zeroS = {};
    const RtAreaLight zeroA = {};
    if (hasSphere && hasArea) {
        const float roulette = Rand(th);
        if (roulette > 0.5f) {
            const uint32_t i = rt_ftou(Rand(th) * (float)(int)s.sphereLights.size());
            SphereLightImportanceSampling(th, hit, ray, i < s.sphereLights.size() ? s.sphereLights[i] : zeroS, func, pdf);
        } else {
            const uint32_t i = rt_ftou(Rand(th) * (float)(int)s.areaLights.size());
            AreaLightImportanceSampling(th, hit, ray, i < s.areaLights.size() ? s.areaLights[i] : zeroA, func, pdf);
        }
    } else {
        if (hasSphere) {
            const uint32_t i = rt_ftou(Rand(th) * (float)(int)s.sphereLights.size());
            SphereLightImportanceSampling(th, hit, ray, i < s.sphereLights.size() ? s.sphereLights[i] : zeroS, func, pdf);
        }
        if (hasArea) {
            const uint32_t i = rt_ftou(Rand(th) * (float)(int)s.areaLights.size());
            AreaLightImportanceSampling(th, hit, ray, i < s.areaLights.size() ? s.areaLights[i] : zeroA, func, pdf);
        }
    }
    if (rt_dot3(ray.direction, hit.normal) < 0.01f) {
        func = rt_splat3(0.0f);
        pdf = -1;
    }
}

// :190-201
f3 LightImportanceSampling(Thread &th, const RayHit &hit, Ray &ray)
{
    f3 func;
    float pdf;
    _LightImportanceSampling(th, hit, ray, func, pdf);
    if (pdf > 0) return rt_divs3(func, pdf);
    return rt_splat3(0.0f);
}

// :204-307
void _BSDFImportanceSampling(Thread &th, const RayHit &hit, Ray &ray, f3 &func, float &pdf)
{
    f3 diffuseReflectionDir, specularReflectionDir;
    const float perceptualRoughness = SmoothnessToPerceptualRoughness(hit.smoothness);
    float roughness = PerceptualRoughnessToRoughness(perceptualRoughness);
    roughness = rt_max(roughness, 0.002f);
    {
        const float theta = rt_acos(rt_sqrt(1 - Rand(th)));
        const float phi = TWO_PI * Rand(th);
        diffuseReflectionDir = Tangent2World(theta, phi, hit.normal);
    }
    {
        const float e = Rand(th);
        const float roughness2 = roughness * roughness;
        const float theta = rt_acos(rt_sqrt((1 - e) / (e * (roughness2 - 1) + 1)));
        const float phi = TWO_PI * Rand(th);
        const f3 microfacetNormal = Tangent2World(theta, phi, hit.normal);
        specularReflectionDir = rt_normalize3(rt_reflect3(ray.direction, microfacetNormal));
    }
    f3 reflectionDir;
    const float diffuseRatio = 0.5f * (1.0f - hit.metallic);
    const float specularRoatio = 1 - diffuseRatio;
    if (Rand(th) < diffuseRatio) reflectionDir = diffuseReflectionDir;
    else reflectionDir = specularReflectionDir;

    if (hit.materialType == 1 || hit.materialType == 3) {
        const bool fromOutside = rt_dot3(ray.direction, hit.normal) < 0;
        const float etai = 1;
        const float etat = hit.ior;
        const float eta = fromOutside ? etai / etat : etat / etai;
        if (hit.materialType == 3) {
            float specularChance = FresnelReflectAmount(!fromOutside ? etat : etai, fromOutside ? etat : etai,
                                                        ray.direction, hit.normal, DIELECTRIC_SPEC, 1.0f);
            // clearCoatColor = hit.clearCoatColor * pow(specularChance,4) * 10 is computed and never used (:271)
            specularChance = rt_pow(specularChance, eta * eta * eta * eta);
            const float pick = (Rand(th) < specularChance) ? 1.0f : 0.0f;
            reflectionDir = rt_lerp3(diffuseReflectionDir, reflectionDir, pick);
        } else if (hit.materialType == 1 && Rand(th) > hit.transparent) {
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
    const f3 lightDir = reflectionDir;
    BRDF(th, viewDir, halfDir, lightDir, hit.albedo, hit.normal, hit.metallic, perceptualRoughness, roughness,
         diffuseRatio, specularRoatio, func, pdf);
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = reflectionDir;
}

// :309-326
void _BSDFImportanceSampling2(const Thread &th, f3 inputDir, f3 outputDir, const RayHit &hit, Ray &ray, f3 &func, float &pdf)
{
    const float perceptualRoughness = SmoothnessToPerceptualRoughness(hit.smoothness);
    float roughness = PerceptualRoughnessToRoughness(perceptualRoughness);
    roughness = rt_max(roughness, 0.002f);
    const float diffuseRatio = 0.5f * (1.0f - hit.metallic);
    const float specularRoatio = 1 - diffuseRatio;
    const f3 viewDir = rt_normalize3(rt_neg3(inputDir));
    const f3 halfDir = rt_normalize3(rt_add3(viewDir, outputDir));
    const f3 lightDir = outputDir;
    BRDF(th, viewDir, halfDir, lightDir, hit.albedo, hit.normal, hit.metallic, perceptualRoughness, roughness,
         diffuseRatio, specularRoatio, func, pdf);
    ray.origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
    ray.direction = outputDir;
}

// :328-339
f3 BSDFImportanceSampling(Thread &th, const RayHit &hit, Ray &ray)
{
    f3 func;
    float pdf;
    _BSDFImportanceSampling(th, hit, ray, func, pdf);
    if (pdf > 0) return rt_divs3(func, pdf);
    return rt_splat3(1.0f);
}

// :342-395
f3 MultipleImportanceSampling(Thread &th, const RayHit &hit, Ray &ray)
{
    f3 lightFunc;
    float lightPdf;
    const f3 inputDir = ray.direction;
    _LightImportanceSampling(th, hit, ray, lightFunc, lightPdf);
    const f3 outputDir = ray.direction;
    const float coin = Rand(th);                       // both operands of && are evaluated
    const bool facing = rt_dot3(hit.normal, outputDir) > 0.01f;
    if (coin > 0.5f && facing) {
        const float lightWeight = 0.5f;
        f3 brdfFunc;
        float brdfPdf;
        const float brdfWeight = 0.5f;
        _BSDFImportanceSampling2(th, inputDir, outputDir, hit, ray, brdfFunc, brdfPdf);
        const f3 func = rt_add3(lightFunc, brdfFunc);
        const float pdf = lightPdf * lightWeight + brdfPdf * brdfWeight;
        if (pdf > 0) return rt_divs3(func, pdf);
        return rt_splat3(0.0f);
    } else {
        ray.direction = inputDir;
        f3 func;
        float pdf;
        _BSDFImportanceSampling(th, hit, ray, func, pdf);
        if (pdf > 0) return rt_divs3(func, pdf);
        return rt_splat3(1.0f);
    }
}

// ------------------------------------------------------------------ a9: Shade.cginc
// :30-38
f3 RotateAroundYInDegrees(f3 dir, float degrees)
{
    const float alpha = degrees * RT_PI / 180.0f;
    float sina, cosa;
    rt_sincos(alpha, &sina, &cosa);
    const float rx = cosa * dir.x + (-sina) * dir.z;
    const float ry = sina * dir.x + cosa * dir.z;
    return rt_mk3(rx, dir.y, ry);
}

// skyboxCube.SampleLevel(sampler_LinearClamp, dir, 0) (Shade.cginc:94) restated
// in software: D3D face selection, bilinear filter inside one face, clamped at
// the face edge (no cross-face filtering).  A direction with a NaN component
// reads texel (0,0) of face +X.  An unbound texture reads 0.
f3 SampleSkybox(const Scene &s, f3 d)
{
    if (s.cubeSize == 0) return rt_splat3(0.0f);
    const int size = (int)s.cubeSize;
    int face = 0;
    float sc = 0, tc = 0, ma = 1;
    bool nan = d.x != d.x || d.y != d.y || d.z != d.z;
    const float ax = rt_abs(d.x), ay = rt_abs(d.y), az = rt_abs(d.z);
    if (!nan) {
        if (ax >= ay && ax >= az) { face = d.x > 0 ? 0 : 1; ma = ax; sc = d.x > 0 ? -d.z : d.z; tc = -d.y; }
        else if (ay >= az)        { face = d.y > 0 ? 2 : 3; ma = ay; sc = d.x; tc = d.y > 0 ? d.z : -d.z; }
        else                      { face = d.z > 0 ? 4 : 5; ma = az; sc = d.z > 0 ? d.x : -d.x; tc = -d.y; }
    }
    const float *img = s.cube.data() + (size_t)face * size * size * 4;
    if (nan || ma == 0.0f) return rt_mk3(img[0], img[1], img[2]);
    const float u = 0.5f * (sc / ma + 1.0f);
    const float v = 0.5f * (tc / ma + 1.0f);
    const float fx = u * (float)size - 0.5f;
    const float fy = v * (float)size - 0.5f;
    const float x0f = rt_floor(fx), y0f = rt_floor(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = std::min(std::max(x0, 0), size - 1); x1 = std::min(std::max(x1, 0), size - 1);
    y0 = std::min(std::max(y0, 0), size - 1); y1 = std::min(std::max(y1, 0), size - 1);
    const float *c00 = img + ((size_t)y0 * size + x0) * 4, *c10 = img + ((size_t)y0 * size + x1) * 4;
    const float *c01 = img + ((size_t)y1 * size + x0) * 4, *c11 = img + ((size_t)y1 * size + x1) * 4;
    const f3 top = rt_lerp3(ld3(c00), ld3(c10), wx);
    const f3 bot = rt_lerp3(ld3(c01), ld3(c11), wx);
    return rt_lerp3(top, bot, wy);
}

f3 FUNCTION_BSDF(Thread &th, const RayHit &hit, Ray &ray)   // Shade.cginc:8-20
{
    switch (th.P->flags & RT_FLAG_SAMPLING_MASK) {
    case RT_SAMPLING_UNIFORM: return UniformSampling(th, hit, ray);
    case RT_SAMPLING_COSINE: return CosineWeightedSampling(th, hit, ray);
    case RT_SAMPLING_LIGHT: return LightImportanceSampling(th, hit, ray);
    case RT_SAMPLING_BSDF: return BSDFImportanceSampling(th, hit, ray);
    default: return MultipleImportanceSampling(th, hit, ray);
    }
}

// Shade.cginc:47-103
f3 Shade(Thread &th, const RayHit &hit, Ray &ray)
{
    if (hit.distance < rt_inf()) {
        if (hit.materialType == 2) return hit.emissionColor;
        else if (hit.materialType == 4) {
            ray.origin = rt_sub3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
            return rt_splat3(0.0f);
        } else {
            ray.energy = rt_mul3(ray.energy, FUNCTION_BSDF(th, hit, ray));
            return rt_splat3(0.0f);
        }
    } else {
        ray.energy = rt_splat3(0.0f);
        const f3 dir = RotateAroundYInDegrees(ray.direction, -th.P->skyboxRotation);
        f3 skyboxColor = SampleSkybox(*th.sc, dir);
        skyboxColor = rt_mk3(rt_linear_to_srgb1(skyboxColor.x), rt_linear_to_srgb1(skyboxColor.y), rt_linear_to_srgb1(skyboxColor.z));
        skyboxColor = rt_scale3(rt_saturate3(skyboxColor), th.P->skyboxExposure);
        skyboxColor = rt_mk3(rt_srgb_to_linear1(skyboxColor.x), rt_srgb_to_linear1(skyboxColor.y), rt_srgb_to_linear1(skyboxColor.z));
        return skyboxColor;
    }
}

// ------------------------------------------------------------------ a10: Shadow.cginc
const float BLUE_NOISE_IN_DISK[16][2] = {   // first NUM_SHADOW_RAYS entries of Shadow.cginc:7-72
    {0.478712f, 0.875764f},  {-0.337956f, -0.793959f}, {-0.955259f, -0.028164f}, {0.864527f, 0.325689f},
    {0.209342f, -0.395657f}, {-0.106779f, 0.672585f},  {0.156213f, 0.235113f},   {-0.413644f, -0.082856f},
    {-0.415667f, 0.323909f}, {0.141896f, -0.939980f},  {0.954932f, -0.182516f},  {-0.766184f, 0.410799f},
    {-0.434912f, -0.458845f}, {0.415242f, -0.078724f}, {0.728335f, -0.491777f},  {-0.058086f, -0.066401f}};

// Shadow.cginc:84-113
f3 SoftShadow(Thread &th, f3 origin, f3 lightDir, float lightRadius)
{
    const float blueNoise = Rand(th);
    const float theta = blueNoise * 2.0f * RT_PI;
    const float cosTheta = rt_cos(theta);
    const float sinTheta = rt_sin(theta);
    float shadowTerm = 0.0f;
    const f3 lightTangent = rt_normalize3(rt_cross3(lightDir, rt_mk3(0.0f, 1.0f, 0.0f)));
    const f3 lightBitangent = rt_normalize3(rt_cross3(lightTangent, lightDir));
    for (int shadowRayIndex = 0; shadowRayIndex < 16; ++shadowRayIndex) {
        const float sx = BLUE_NOISE_IN_DISK[shadowRayIndex][0], sy = BLUE_NOISE_IN_DISK[shadowRayIndex][1];
        float dx = sx * cosTheta - sy * sinTheta;
        float dy = sx * sinTheta + sy * cosTheta;
        dx *= lightRadius;
        dy *= lightRadius;
        const f3 shadowRayDir = rt_normalize3(rt_add3(rt_add3(lightDir, rt_scale3(lightTangent, dx)), rt_scale3(lightBitangent, dy)));
        const Ray shadowRay = CreateRay(origin, shadowRayDir);
        const RayHit shadowHit = BVHRayTrace(th, shadowRay);
        if (shadowHit.castShadow > 0)
            shadowTerm = rt_lerp(shadowTerm, (shadowHit.distance == rt_inf()) ? 0.0f : 1.0f, 1.0f / (float)(shadowRayIndex + 1));
    }
    const float *sp = th.P->shadowParameter;
    return rt_lerp3(rt_splat3(1.0f), rt_mk3(sp[0], sp[1], sp[2]), sp[3] * shadowTerm);
}

// ------------------------------------------------------------------ kernels
// RayTracing.compute:44-91
void CSMain(const Scene &s, const RtParams &P, uint32_t idx, uint32_t idy, float out[4], TraceCounters *cnt)
{
    Thread th;
    th.sc = &s;
    th.P = &P;
    th.cnt = cnt;
    rt_rng_init(&th.rng, idx, idy, P.seed);
    const float r0 = Rand(th), r1 = Rand(th);
    const float uvx = (((float)idx + r0) - 0.5f) / (float)P.width * 2 - 1;
    const float uvy = (((float)idy + r1) - 0.5f) / (float)P.height * 2 - 1;
    Ray ray = (P.flags & RT_FLAG_DOF_PRIMARY) ? CreateDoFCameraRay(th, uvx, uvy) : CreateCameraRay(th, uvx, uvy);
    f3 col = rt_splat3(0.0f);
    for (int i = 0; i < RT_MAX_BOUNCE; i++) {
        const RayHit hit = (P.flags & RT_FLAG_BRUTE_FORCE) ? BruteForceRayTrace(th, ray) : BVHRayTrace(th, ray);   // :73-77
        const f3 energyBefore = ray.energy;           // ray.energy * Shade(...): left operand read first
        const f3 sh = Shade(th, hit, ray);
        col = rt_add3(col, rt_mul3(energyBefore, sh));
        if (!rt_any3(ray.energy)) break;
    }
    out[0] = col.x; out[1] = col.y; out[2] = col.z; out[3] = 0.0f;
}

// RayTracing.compute:96-131
void ShadowMapKernel(const Scene &s, const RtParams &P, uint32_t idx, uint32_t idy, float out[4], TraceCounters *cnt)
{
    Thread th;
    th.sc = &s;
    th.P = &P;
    th.cnt = cnt;
    rt_rng_init(&th.rng, idx, idy, P.seed);
    const float r0 = Rand(th), r1 = Rand(th);
    const float uvx = (((float)idx + r0) - 0.5f) / (float)P.width * 2 - 1;
    const float uvy = (((float)idy + r1) - 0.5f) / (float)P.height * 2 - 1;
    const Ray ray = (P.flags & RT_FLAG_DEPTH_OF_FIELD) ? CreateDoFCameraRay(th, uvx, uvy) : CreateCameraRay(th, uvx, uvy);
    const RayHit hit = BVHRayTrace(th, ray);
    if (hit.distance < rt_inf()) {
        const f3 lightDir = rt_mk3(-P.lightParameter[0], -P.lightParameter[1], -P.lightParameter[2]);
        const f3 shadow = SoftShadow(th, rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS)), lightDir, P.lightParameter[3]);
        out[0] = shadow.x; out[1] = shadow.y; out[2] = shadow.z;
        out[3] = hit.materialType == 4 ? 0.0f : 1.0f;
    } else {
        out[0] = 1; out[1] = 1; out[2] = 1; out[3] = 0;
    }
}

} // namespace

// ======================================================================= C API
extern "C" {

void *orc_scene_create(const RtMeshVertex *v, uint32_t nV, const uint32_t *idx, uint32_t nIdx,
                       const uint32_t *matIdx, const int32_t *shadow, uint32_t nTri)
{
    Scene *s = new Scene;
    s->verts.assign(v, v + nV);
    s->idx.assign(idx, idx + nIdx);
    s->matIdx.assign(matIdx, matIdx + nTri);
    s->shadow.assign(shadow, shadow + 2 * (size_t)nTri);
    s->N = nTri;
    return s;
}
void orc_scene_destroy(void *h) { delete (Scene *)h; }
void orc_set_materials(void *h, const RtMaterialData *m, uint32_t n) { ((Scene *)h)->mats.assign(m, m + n); }
void orc_set_lights(void *h, const RtSphereLight *sl, uint32_t ns, const RtAreaLight *al, uint32_t na,
                    const RtDiscLight *dl, uint32_t nd)
{
    Scene *s = (Scene *)h;
    s->sphereLights.assign(sl, sl + ns);
    s->areaLights.assign(al, al + na);
    s->discLights.assign(dl, dl + nd);
}
// MeshCollector.Init (MeshCollector.cs:170-200)
void orc_set_primitives(void *h, const RtPlane *planes, uint32_t nPlanes, const RtSphere *spheres, uint32_t nSpheres, const RtCube *cubes,
                        uint32_t nCubes, const RtCustomMesh *meshes, uint32_t nMeshes, const float *vertices3, const float *normals3,
                        uint32_t nVertices, const int32_t *indices, uint32_t nIndices)
{
    Scene *s = (Scene *)h;
    s->planeBuffer.assign(planes, planes + nPlanes);
    s->sphereBuffer.assign(spheres, spheres + nSpheres);
    s->cubeBuffer.assign(cubes, cubes + nCubes);
    s->customMeshBuffer.assign(meshes, meshes + nMeshes);
    s->vertexBuffer.resize(nVertices);
    s->normalBuffer.resize(nVertices);
    for (uint32_t i = 0; i < nVertices; ++i) {
        s->vertexBuffer[i] = ld3(vertices3 + 3 * (size_t)i);
        s->normalBuffer[i] = ld3(normals3 + 3 * (size_t)i);
    }
    s->indexBuffer.assign(indices, indices + nIndices);
}

// BruteForceRayTrace for caller rays: distance (inf where Shade sees a miss), last writer, normal
void orc_trace_primitives(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_prim, float *out_normal)
{
    const Scene &s = *(Scene *)h;
    Thread th;
    th.sc = &s;
    th.P = nullptr;
    th.cnt = nullptr;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Ray ray;
        ray.origin = ld3(o + 3 * i);
        ray.direction = ld3(d + 3 * i);
        ray.energy = rt_splat3(1.0f);
        const RayHit hit = BruteForceRayTrace(th, ray);
        const bool isHit = hit.distance < rt_inf();
        out_t[i] = isHit ? hit.distance : rt_inf();
        out_prim[i] = isHit ? hit.triangle : 0xFFFFFFFFu;
        out_normal[3 * i] = isHit ? hit.normal.x : 0.0f;
        out_normal[3 * i + 1] = isHit ? hit.normal.y : 0.0f;
        out_normal[3 * i + 2] = isHit ? hit.normal.z : 0.0f;
    }
}

void orc_set_skybox(void *h, const float *faces, uint32_t size)
{
    Scene *s = (Scene *)h;
    s->cubeSize = size;
    s->cube.assign(faces, faces + (size_t)6 * size * size * 4);
}
void orc_build(void *h) { Build(*(Scene *)h); }

void orc_readback_bvh(void *h, uint32_t *morton, uint32_t *sortedIdx, RtAABB *triAABB, RtTriangle *tris,
                      RtInternalNode *inodes, RtLeafNode *leaves, RtAABB *bvhData)
{
    Scene *s = (Scene *)h;
    const size_t N = s->N;
    if (morton) memcpy(morton, s->morton.data(), N * 4);
    if (sortedIdx) memcpy(sortedIdx, s->sortedIdx.data(), N * 4);
    if (triAABB) memcpy(triAABB, s->triAABB.data(), N * sizeof(RtAABB));
    if (tris) memcpy(tris, s->tris.data(), N * sizeof(RtTriangle));
    if (inodes && N > 1) memcpy(inodes, s->inodes.data(), (N - 1) * sizeof(RtInternalNode));
    if (leaves) memcpy(leaves, s->leaves.data(), N * sizeof(RtLeafNode));
    if (bvhData && N > 1) memcpy(bvhData, s->bvhData.data(), (N - 1) * sizeof(RtAABB));
}
void orc_readback_sort_stages(void *h, uint32_t *unsorted, uint32_t *sortedRaw)
{
    Scene *s = (Scene *)h;
    if (unsorted) memcpy(unsorted, s->mortonUnsorted.data(), (size_t)s->N * 4);
    if (sortedRaw) memcpy(sortedRaw, s->mortonSortedRaw.data(), (size_t)s->N * 4);
}

// Stand-alone stages on caller data, for unit checks.
void orc_radix_sort(uint32_t *keys, uint32_t *vals, uint32_t n)
{
    std::vector<uint32_t> k(keys, keys + n), v(vals, vals + n);
    RadixSort(k, v);
    memcpy(keys, k.data(), (size_t)n * 4);
    memcpy(vals, v.data(), (size_t)n * 4);
}
void orc_distribute_morton(uint32_t *keys, uint32_t n)
{
    std::vector<uint32_t> k(keys, keys + n);
    DistributeMortonCode(k, n);
    memcpy(keys, k.data(), (size_t)n * 4);
}
uint32_t orc_morton_code(float x, float y, float z) { return MortonCode(rt_mk3(x, y, z)); }
void orc_tree_from_codes(const uint32_t *codes, uint32_t n, RtInternalNode *inodes, RtLeafNode *leaves)
{
    Scene s;
    s.N = n;
    s.morton.assign(codes, codes + n);
    TreeConstructor(s);
    if (n > 1) memcpy(inodes, s.inodes.data(), (size_t)(n - 1) * sizeof(RtInternalNode));
    memcpy(leaves, s.leaves.data(), (size_t)n * sizeof(RtLeafNode));
}

void orc_set_prune_scale(float scale) { g_pruneScale = scale > 0.0f ? scale : rt_inf(); }

static uint32_t *g_nodeVisits = nullptr;
// Analysis aid: while set, orc_trace adds 1 to visits[node] for every internal node whose box a ray passes.
void orc_set_node_visit_histogram(uint32_t *visits) { g_nodeVisits = visits; }

// First-hit queries.  counters (5 x u64: rays, internal, leaf, tri, hits) may be NULL.
void orc_trace(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_tri,
               float *out_u, float *out_v, uint64_t *counters)
{
    const Scene &s = *(Scene *)h;
    TraceCounters total = {0, 0, 0, 0, 0};
#pragma omp parallel
    {
        TraceCounters c = {0, 0, 0, 0, 0};
        c.nodeVisits = g_nodeVisits;
        Thread th;
        th.sc = &s;
        th.P = nullptr;
        th.cnt = &c;
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)n; ++i) {
            Ray r = CreateRay(ld3(o + 3 * i), ld3(d + 3 * i));
            RayHit hit = BVHRayTrace(th, r);
            if (out_t) out_t[i] = hit.distance;
            if (out_tri) out_tri[i] = hit.triangle;
            if (out_u) out_u[i] = hit.u;
            if (out_v) out_v[i] = hit.v;
        }
#pragma omp critical
        {
            total.rays += c.rays; total.internalVisits += c.internalVisits; total.leafVisits += c.leafVisits;
            total.triTests += c.triTests; total.hits += c.hits;
        }
    }
    if (counters) {
        counters[0] = total.rays; counters[1] = total.internalVisits; counters[2] = total.leafVisits;
        counters[3] = total.triTests; counters[4] = total.hits;
    }
}

// The same queries through the grid walk (IntersectTriangleGrid) or, wide != 0, the walk that takes two levels of
// the tree per step (IntersectTriangleWide).
static void trace_walk(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_tri,
                       float *out_u, float *out_v, uint64_t *counters, int wide)
{
    const Scene &s = *(Scene *)h;
    TraceCounters total = {0, 0, 0, 0, 0};
#pragma omp parallel
    {
        TraceCounters c = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)n; ++i) {
            Ray r = CreateRay(ld3(o + 3 * i), ld3(d + 3 * i));
            RayHit hit = CreateRayHit();
            c.rays++;
            if (wide) IntersectTriangleWide(s, r, hit, &c);
            else IntersectTriangleGrid(s, r, hit, &c);
            if (hit.distance < rt_inf()) c.hits++;
            if (out_t) out_t[i] = hit.distance;
            if (out_tri) out_tri[i] = hit.triangle;
            if (out_u) out_u[i] = hit.u;
            if (out_v) out_v[i] = hit.v;
        }
#pragma omp critical
        {
            total.rays += c.rays; total.internalVisits += c.internalVisits; total.leafVisits += c.leafVisits;
            total.triTests += c.triTests; total.hits += c.hits;
        }
    }
    if (counters) {
        counters[0] = total.rays; counters[1] = total.internalVisits; counters[2] = total.leafVisits;
        counters[3] = total.triTests; counters[4] = total.hits;
    }
}

void orc_trace_grid(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_tri,
                    float *out_u, float *out_v, uint64_t *counters)
{
    trace_walk(h, o, d, n, out_t, out_tri, out_u, out_v, counters, 0);
}
void orc_trace_wide(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_tri,
                    float *out_u, float *out_v, uint64_t *counters)
{
    trace_walk(h, o, d, n, out_t, out_tri, out_u, out_v, counters, 1);
}

// 1 when every box fitted the grid; grid6 = org[3], step[3]; qboxes = 6 words per internal node (may be NULL).
// The library's traversal tree as rebuilt here: wire structs of the second tree (lengths as orc_readback_bvh)
// and every triangle's place in the reference's visiting order.  Any pointer may be null.
void orc_readback_walk(void *h, uint32_t *sortedIdx, RtInternalNode *inodes, RtLeafNode *leaves, RtAABB *bvhData, uint32_t *visitRank)
{
    const Scene &s = *(Scene *)h;
    if (visitRank) memcpy(visitRank, s.visitRank.data(), s.visitRank.size() * 4);
    if (!s.walk) return;
    const Scene &w = *s.walk;
    if (sortedIdx) memcpy(sortedIdx, w.sortedIdx.data(), w.sortedIdx.size() * 4);
    if (inodes) memcpy(inodes, w.inodes.data(), w.inodes.size() * sizeof(RtInternalNode));
    if (leaves) memcpy(leaves, w.leaves.data(), w.leaves.size() * sizeof(RtLeafNode));
    if (bvhData) memcpy(bvhData, w.bvhData.data(), w.bvhData.size() * sizeof(RtAABB));
}

int orc_grid(void *h, float *grid6, uint32_t *qboxes)
{
    const Scene &s = *(Scene *)h;
    if (s.N >= 2) {
        if (grid6) { memcpy(grid6, s.grid.org, 12); memcpy(grid6 + 3, s.grid.step, 12); }
        if (qboxes) memcpy(qboxes, s.qboxes.data(), s.qboxes.size() * 4);
    }
    return s.gridOk ? 1 : 0;
}

// Brute force over every triangle in original order, no boxes: the closest
// Moller-Trumbore hit, ties to the lowest triangle index.
void orc_trace_brute(void *h, const float *o, const float *d, uint32_t n, float *out_t, uint32_t *out_tri)
{
    const Scene &s = *(Scene *)h;
#pragma omp parallel for schedule(dynamic, 64)
    for (long long i = 0; i < (long long)n; ++i) {
        Ray r = CreateRay(ld3(o + 3 * i), ld3(d + 3 * i));
        float best = rt_inf();
        uint32_t bt = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < s.N; ++k) {
            const RtTriangle &tri = s.tris[k];
            float t = 0, u = 0, v = 0;
            if (IntersectTriangle_MT97(r, ld3(tri.point0), ld3(tri.point1), ld3(tri.point2), t, u, v) && t > 0 && t < best) {
                best = t;
                bt = k;
            }
        }
        out_t[i] = best;
        out_tri[i] = bt;
    }
}

// One CSMain frame over the whole image (dst = width*height*4 floats, row 0 = id.y 0).
void orc_render(void *h, const RtParams *P, float *dst, uint64_t *counters)
{
    const Scene &s = *(Scene *)h;
    TraceCounters total = {0, 0, 0, 0, 0};
#pragma omp parallel
    {
        TraceCounters c = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < P->height; ++y)
            for (int x = 0; x < P->width; ++x)
                CSMain(s, *P, (uint32_t)x, (uint32_t)y, dst + ((size_t)y * P->width + x) * 4, &c);
#pragma omp critical
        {
            total.rays += c.rays; total.internalVisits += c.internalVisits; total.leafVisits += c.leafVisits;
            total.triTests += c.triTests; total.hits += c.hits;
        }
    }
    if (counters) {
        counters[0] = total.rays; counters[1] = total.internalVisits; counters[2] = total.leafVisits;
        counters[3] = total.triTests; counters[4] = total.hits;
    }
}

// CSMain (and, shadow != NULL, the ShadowMap kernel) for the pixels of a rectangle of the frame only: dst / shadow =
// w*h*4 floats, row 0 = id.y y0.  Pixel coordinates stay the frame's, so the values are those of the whole-frame run.
void orc_render_region(void *h, const RtParams *P, int x0, int y0, int w, int hgt, float *dst, float *shadow)
{
    const Scene &s = *(Scene *)h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < hgt; ++y)
        for (int x = 0; x < w; ++x) {
            CSMain(s, *P, (uint32_t)(x0 + x), (uint32_t)(y0 + y), dst + ((size_t)y * w + x) * 4, nullptr);
            if (shadow) ShadowMapKernel(s, *P, (uint32_t)(x0 + x), (uint32_t)(y0 + y), shadow + ((size_t)y * w + x) * 4, nullptr);
        }
}

void orc_shadowmap(void *h, const RtParams *P, float *dst)
{
    const Scene &s = *(Scene *)h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < P->height; ++y)
        for (int x = 0; x < P->width; ++x)
            ShadowMapKernel(s, *P, (uint32_t)x, (uint32_t)y, dst + ((size_t)y * P->width + x) * 4, nullptr);
}

// The same kernel with the traversal work counted: 1 primary ray per pixel and 16 soft-shadow rays per pixel
// whose primary ray hit (rays, internalVisits, leafVisits, triTests, hits).
void orc_shadowmap_counted(void *h, const RtParams *P, float *dst, uint64_t *counters)
{
    const Scene &s = *(Scene *)h;
    TraceCounters total = {0, 0, 0, 0, 0};
#pragma omp parallel
    {
        TraceCounters c = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < P->height; ++y)
            for (int x = 0; x < P->width; ++x)
                ShadowMapKernel(s, *P, (uint32_t)x, (uint32_t)y, dst + ((size_t)y * P->width + x) * 4, &c);
#pragma omp critical
        {
            total.rays += c.rays; total.internalVisits += c.internalVisits; total.leafVisits += c.leafVisits;
            total.triTests += c.triTests; total.hits += c.hits;
        }
    }
    counters[0] = total.rays; counters[1] = total.internalVisits; counters[2] = total.leafVisits;
    counters[3] = total.triTests; counters[4] = total.hits;
}

// AddShader.shader:12,47-79 with Blend SrcAlpha OneMinusSrcAlpha on a float target.
void orc_accumulate(float *converged, const float *rt, const float *shadowMap, uint32_t samplePrePixel,
                    uint64_t nPixels, int pass)
{
    const float a = 1.0f / ((float)samplePrePixel + 1.0f);
    for (uint64_t i = 0; i < nPixels; ++i) {
        float c[3] = {rt[i * 4 + 0], rt[i * 4 + 1], rt[i * 4 + 2]};
        if (pass == 1)
            for (int k = 0; k < 3; ++k) c[k] *= shadowMap[i * 4 + k];
        for (int k = 0; k < 3; ++k)
            converged[i * 4 + k] = c[k] * a + converged[i * 4 + k] * (1.0f - a);
        converged[i * 4 + 3] = a * a + converged[i * 4 + 3] * (1.0f - a);
    }
}

// Known-answer helpers.
void orc_rand_sequence(uint32_t px, uint32_t py, float seed, uint32_t n, float *out, uint32_t *out_state)
{
    rt_rng r;
    rt_rng_init(&r, px, py, seed);
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = rt_rand(&r);
        if (out_state) out_state[i] = r.state;
    }
}
// x = pixel(2), hit.position(3), hit.normal(3), hit.albedo(3), light.position(3), light.radius; out = origin, direction, func, pdf
void orc_disc_light(const float *x, float seed, float *out)
{
    Thread th;
    th.sc = nullptr;
    th.P = nullptr;
    th.cnt = nullptr;
    rt_rng_init(&th.rng, (uint32_t)x[0], (uint32_t)x[1], seed);
    RayHit hit = CreateRayHit();
    hit.position = ld3(x + 2);
    hit.normal = ld3(x + 5);
    hit.albedo = ld3(x + 8);
    RtDiscLight light;
    memset(&light, 0, sizeof light);
    memcpy(light.position, x + 11, 12);
    light.radius = x[14];
    Ray ray;
    ray.origin = ray.direction = rt_splat3(0.0f);
    ray.energy = rt_splat3(1.0f);
    f3 func;
    float pdf;
    DiscLightImportanceSampling(th, hit, ray, light, func, pdf);
    const float o[10] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, func.x, func.y, func.z, pdf};
    memcpy(out, o, sizeof o);
}
// which: 0 sin, 1 cos, 2 acos, 3 pow(x,y), 4 srgb->linear, 5 linear->srgb, 6 ftou (as float bits of the uint)
void orc_math(int which, const float *x, const float *y, uint32_t n, float *out)
{
    for (uint32_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: out[i] = rt_sin(x[i]); break;
        case 1: out[i] = rt_cos(x[i]); break;
        case 2: out[i] = rt_acos(x[i]); break;
        case 3: out[i] = rt_pow(x[i], y[i]); break;
        case 4: out[i] = rt_srgb_to_linear1(x[i]); break;
        case 5: out[i] = rt_linear_to_srgb1(x[i]); break;
        default: out[i] = rt_u2f_bits(rt_ftou(x[i])); break;
        }
    }
}
// Shading-function table: evaluates BRDF() for n (view, light, normal) triples
// with one material, for per-function checks.  in = n x 9 floats, out = n x 4 (func.rgb, pdf).
void orc_brdf_table(const RtParams *P, const RtMaterialData *m, const float *in, uint32_t n, float *out)
{
    Scene s;
    Thread th;
    th.sc = &s;
    th.P = P;
    th.cnt = nullptr;
    for (uint32_t i = 0; i < n; ++i) {
        const f3 viewDir = ld3(in + 9 * i), lightDir = ld3(in + 9 * i + 3), normal = ld3(in + 9 * i + 6);
        const f3 halfDir = rt_normalize3(rt_add3(viewDir, lightDir));
        const float pr = SmoothnessToPerceptualRoughness(m->smoothness);
        const float rough = rt_max(PerceptualRoughnessToRoughness(pr), 0.002f);
        const float diffuseRatio = 0.5f * (1.0f - m->metallic);
        f3 func;
        float pdf;
        BRDF(th, viewDir, halfDir, lightDir, ld3(m->albedo), normal, m->metallic, pr, rough, diffuseRatio, 1 - diffuseRatio, func, pdf);
        out[4 * i + 0] = func.x; out[4 * i + 1] = func.y; out[4 * i + 2] = func.z; out[4 * i + 3] = pdf;
    }
}
int orc_num_threads(void)
{
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
#endif
    return n;
}

} // extern "C"
