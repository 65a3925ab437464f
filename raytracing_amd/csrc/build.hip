// build.hip -- LBVH construction on gfx950: per-triangle boxes + Morton codes,
// LDS radix sort, duplicate-key spreading, Karras tree, bottom-up box fit and
// packing into the 64-byte traversal nodes.
//
// Reference path being replaced (results are bit-identical on the same input):
//   Assets/Shaders/BVH/MeshData.compute:62-131          (k_calculate)
//   Assets/Shaders/BVH/LocalRadixSort.compute, Scan.compute, GlobalRadixSort.compute,
//   Assets/Scripts/BVH/ComputeBufferSorter.cs:105-131    (k_sort_*: stable LSD, 4 x 8 bit)
//   Assets/Scripts/BVH/MeshBufferContainer.cs:92-107     (k_scan_* in dedup mode, on the GPU)
//   Assets/Shaders/BVH/BVH.compute:94-142                (k_tree)
//   Assets/Shaders/BVH/BVH.compute:164-211               (k_fit, level-synchronous, no atomics)
#include "rt_internal.h"
#include "../../include/rt_finekey.h"
#include "../../include/rt_qh.h"
#include "../../include/rt_rotate.h"

#define WAVE 64
#define SORT_THREADS 256
#define SORT_ITEMS 16
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)
#define SCAN_THREADS 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)
#define FIT_UNDONE 0xFFFFFFFFu

// ------------------------------------------------------------------ calculate
__device__ inline uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ inline uint32_t morton_code(rt_f3 v)
{
    v.x = rt_min(rt_max(v.x * 1024.0f, 0.0f), 1023.0f);
    v.y = rt_min(rt_max(v.y * 1024.0f, 0.0f), 1023.0f);
    v.z = rt_min(rt_max(v.z * 1024.0f, 0.0f), 1023.0f);
    const uint32_t xx = expand_bits(rt_ftou(v.x));
    const uint32_t yy = expand_bits(rt_ftou(v.y));
    const uint32_t zz = expand_bits(rt_ftou(v.z));
    return xx * 4 + yy * 2 + zz;
}

__global__ __launch_bounds__(256) void k_calculate(const uint32_t *__restrict__ idx, const float *__restrict__ verts,
                                                    uint32_t N, float4 *__restrict__ triAABB,
                                                    uint32_t *__restrict__ morton, uint32_t *__restrict__ triIndex)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= N) return;
    const uint32_t i0 = idx[id * 3 + 0], i1 = idx[id * 3 + 1], i2 = idx[id * 3 + 2];
    const float *a = verts + (size_t)i0 * 11, *b = verts + (size_t)i1 * 11, *c = verts + (size_t)i2 * 11;
    const rt_f3 p0 = rt_mk3(a[0], a[1], a[2]), p1 = rt_mk3(b[0], b[1], b[2]), p2 = rt_mk3(c[0], c[1], c[2]);
    const rt_f3 mn = rt_mk3(rt_min(rt_min(p0.x, p1.x), p2.x) - 0.001f, rt_min(rt_min(p0.y, p1.y), p2.y) - 0.001f,
                            rt_min(rt_min(p0.z, p1.z), p2.z) - 0.001f);
    const rt_f3 mx = rt_mk3(rt_max(rt_max(p0.x, p1.x), p2.x) + 0.001f, rt_max(rt_max(p0.y, p1.y), p2.y) + 0.001f,
                            rt_max(rt_max(p0.z, p1.z), p2.z) + 0.001f);
    triAABB[id * 2 + 0] = make_float4(mn.x, mn.y, mn.z, 0.0f);
    triAABB[id * 2 + 1] = make_float4(mx.x, mx.y, mx.z, 0.0f);
    // Fixed +-125 world box (MeshBufferContainer.cs:9-14,73-77), not the mesh bounds.
    const rt_f3 encMin = rt_mk3(-125.0f, -125.0f, -125.0f), encMax = rt_mk3(125.0f, 125.0f, 125.0f);
    const rt_f3 center = rt_scale3(rt_add3(mn, mx), 0.5f);
    const rt_f3 normalized = rt_div3(rt_sub3(center, encMin), rt_sub3(encMax, encMin));
    morton[id] = morton_code(normalized);
    triIndex[id] = id;
}

// ------------------------------------------------------------------ block scan helper
// Exclusive scan of one value per thread over a block of up to 1024 threads.
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *waveSums /* LDS, >= 17 */, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, WAVE);
        if (lane >= (uint32_t)off) inc += t;
    }
    if (lane == 63u) waveSums[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t t = waveSums[w];
            waveSums[w] = run;
            run += t;
        }
        waveSums[16] = run;
    }
    __syncthreads();
    const uint32_t res = waveSums[wave] + inc - v;
    *total = waveSums[16];
    __syncthreads();
    return res;
}

// ------------------------------------------------------------------ scan (3 kernels)
// MODE 0: out[i] = sum_{j<i} in[j]            (exclusive, plain)
// MODE 1: f(0) = 0, f(i) = max(in[i]-in[i-1], 1); out[i] = sum_{j<=i} f(j)
//         = DistributeMortonCode (MeshBufferContainer.cs:92-107) in u32 arithmetic.
template <int MODE>
__device__ inline uint32_t scan_value(const uint32_t *__restrict__ in, uint32_t i)
{
    if (MODE == 0) return in[i];
    if (i == 0) return 0u;
    const uint32_t d = in[i] - in[i - 1];
    return d > 1u ? d : 1u;
}

template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t M,
                                                               uint32_t *__restrict__ blockSums)
{
    __shared__ uint32_t ws[17];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * SCAN_THREADS + threadIdx.x;
        if (i < M) s += scan_value<MODE>(in, i);
    }
    uint32_t total;
    block_exclusive_scan(s, ws, &total);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ sums, uint32_t n)
{
    __shared__ uint32_t ws[17];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, ws, &total);
        if (i < n) sums[i] = carry + ex;
        carry += total;
    }
}

template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              uint32_t M, const uint32_t *__restrict__ blockSums)
{
    __shared__ uint32_t ws[17];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;  // blocked: 16 consecutive per thread
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k;
        v[k] = i < M ? scan_value<MODE>(in, i) : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, ws, &total) + blockSums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k;
        if (MODE == 0) {
            if (i < M) out[i] = run;
            run += v[k];
        } else {
            run += v[k];
            if (i < M) out[i] = run;
        }
    }
}

template <int MODE>
static void launch_scan(RtContext *ctx, const uint32_t *in, uint32_t *out, uint32_t M, uint32_t *blockSums)
{
    const uint32_t nb = (M + SCAN_TILE - 1) / SCAN_TILE;
    rt_timer_begin(ctx, MODE == 0 ? "k_scan" : "k_dedup");
    hipLaunchKernelGGL((k_scan_reduce<MODE>), dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, in, M, blockSums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, blockSums, nb);
    hipLaunchKernelGGL((k_scan_apply<MODE>), dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, M, blockSums);
    rt_timer_end(ctx);
}

// ------------------------------------------------------------------ radix sort
// Tile = 4096 keys per 256-thread block.  Wave w owns keys [w*1024,(w+1)*1024)
// of the tile and walks them in 16 rounds of 64 consecutive keys, so that
// (wave, round, lane) order == memory order == the order stability refers to.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t *__restrict__ keys, uint32_t n, int shift,
                                                             uint32_t *__restrict__ blockHist, uint32_t nb)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockHist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];   // digit-major
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const uint32_t *__restrict__ keysIn,
                                                                const uint32_t *__restrict__ valsIn,
                                                                uint32_t *__restrict__ keysOut,
                                                                uint32_t *__restrict__ valsOut, uint32_t n, int shift,
                                                                const uint32_t *__restrict__ scanned, uint32_t nb)
{
    __shared__ uint32_t waveCount[4][256];
    __shared__ uint32_t base[256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    for (uint32_t i = tid; i < 1024; i += SORT_THREADS) (&waveCount[0][0])[i] = 0;
    __syncthreads();

    uint32_t key[SORT_ITEMS], rank[SORT_ITEMS];
    const uint32_t tileBase = blockIdx.x * SORT_TILE + wave * (SORT_ITEMS * WAVE);
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = tileBase + r * WAVE + lane;
        const bool valid = i < n;
        const uint32_t k = valid ? keysIn[i] : 0xFFFFFFFFu;
        const uint32_t d = (k >> shift) & 255u;
        // lanes of this wave holding the same digit (wave64 ballots, one per bit)
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t rnk = __popcll(mask & lt);
        const uint32_t cnt = __popcll(mask);
        uint32_t prev = 0;
        if (valid && rnk == 0) prev = atomicAdd(&waveCount[wave][d], cnt);   // one leader per digit group
        const int leader = valid ? (__ffsll((long long)mask) - 1) : (int)lane;
        prev = __shfl(prev, leader, WAVE);
        rank[r] = prev + rnk;
        key[r] = k;
    }
    __syncthreads();
    {   // exclusive prefix over the 4 waves per digit + global base of (digit, block)
        const uint32_t d = tid;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t t = waveCount[w][d];
            waveCount[w][d] = run;
            run += t;
        }
        base[d] = scanned[d * nb + blockIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = tileBase + r * WAVE + lane;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t dst = base[d] + waveCount[wave][d] + rank[r];
            keysOut[dst] = key[r];
            valsOut[dst] = valsIn[i];
        }
    }
}

// ------------------------------------------------------------------ tree (Karras 2012)
__device__ inline int delta_codes(const uint32_t *__restrict__ codes, int x, int y, int numObjects)
{
    if (x >= 0 && x <= numObjects - 1 && y >= 0 && y <= numObjects - 1) {
        const uint32_t v = codes[x] ^ codes[y];
        return v == 0 ? 32 : __clz((int)v);
    }
    return -1;
}

// rangeSize (may be null): number of leaves under each node = length of its key range
__global__ __launch_bounds__(256) void k_tree(const uint32_t *__restrict__ codes, uint32_t N, uint32_t *__restrict__ left,
                                               uint32_t *__restrict__ right, uint32_t *__restrict__ parent,
                                               uint32_t *__restrict__ leafParent, uint32_t *__restrict__ rangeSize)
{
    const uint32_t threadId = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadId >= N - 1) return;
    const int numObjects = (int)N, idx = (int)threadId;
    // DetermineRange (BVH.compute:35-52); uint products wrap modulo 2^32 as in the shader
    const int dd = delta_codes(codes, idx, idx + 1, numObjects) - delta_codes(codes, idx, idx - 1, numObjects);
    const int d = dd > 0 ? 1 : (dd < 0 ? -1 : 0);
    const int dmin = delta_codes(codes, idx, idx - d, numObjects);
    uint32_t lmax = 2;
    while (delta_codes(codes, idx, (int)((uint32_t)idx + lmax * (uint32_t)d), numObjects) > dmin) lmax *= 2;
    int l = 0;
    for (uint32_t t = lmax / 2; t >= 1; t /= 2)
        if (delta_codes(codes, idx, (int)((uint32_t)idx + ((uint32_t)l + t) * (uint32_t)d), numObjects) > dmin) l += (int)t;
    const int j = idx + l * d;
    const int first = idx < j ? idx : j, last = idx < j ? j : idx;
    if (rangeSize) rangeSize[threadId] = (uint32_t)(last - first + 1);
    // FindSplit (BVH.compute:54-92)
    int split;
    {
        const uint32_t firstCode = codes[first], lastCode = codes[last];
        if (firstCode == lastCode) {
            split = (first + last) >> 1;
        } else {
            const int commonPrefix = __clz((int)(firstCode ^ lastCode));
            split = first;
            int step = last - first;
            do {
                step = (step + 1) >> 1;
                const int newSplit = split + step;
                if (newSplit < last) {
                    const uint32_t x = firstCode ^ codes[newSplit];
                    const int splitPrefix = x == 0 ? 32 : __clz((int)x);
                    if (splitPrefix > commonPrefix) split = newSplit;
                }
            } while (step > 1);
        }
    }
    if (split == first) {
        leafParent[split] = threadId;
        left[threadId] = (uint32_t)split | RT_LEAF_BIT;
    } else {
        parent[split] = threadId;
        left[threadId] = (uint32_t)split;
    }
    if (split + 1 == last) {
        leafParent[split + 1] = threadId;
        right[threadId] = (uint32_t)(split + 1) | RT_LEAF_BIT;
    } else {
        parent[split + 1] = threadId;
        right[threadId] = (uint32_t)(split + 1);
    }
}

// ------------------------------------------------------------------ bottom-up fit
// One launch per tree level: a node is computed in launch `iter` when both of
// its internal children were completed in an EARLIER launch (their `done` word
// is < iter), so every read of a child box is ordered by a kernel boundary and
// no atomics or fences are needed.  The union itself is min/max only, hence
// identical to the reference's atomic-flag walk whatever the order.
// (`list`: the nodes still to do, nullptr = all n of them; see compact_undone)
__global__ __launch_bounds__(256) void k_fit(uint32_t nAll, const uint32_t *__restrict__ list, const uint32_t *__restrict__ listCount, uint32_t iter, const uint32_t *__restrict__ left,
                                              const uint32_t *__restrict__ right, const uint32_t *__restrict__ sortedIdx,
                                              const float4 *__restrict__ triAABB, float4 *bvhData, uint32_t *done,
                                              float4 *__restrict__ nodes, const uint32_t *__restrict__ parent)
{
    const uint32_t n = list ? *listCount : nAll;          // the list's length stays on the device
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint32_t i = list ? list[k] : k;
    if (done[i] != FIT_UNDONE) continue;
    const uint32_t l = left[i], r = right[i];
    const bool lLeaf = l & RT_LEAF_BIT, rLeaf = r & RT_LEAF_BIT;
    const uint32_t li = l & ~RT_LEAF_BIT, ri = r & ~RT_LEAF_BIT;
    if (!lLeaf && done[li] >= iter) continue;
    if (!rLeaf && done[ri] >= iter) continue;
    float4 lmn, lmx, rmn, rmx;
    if (lLeaf) { const uint32_t t = sortedIdx[li]; lmn = triAABB[t * 2]; lmx = triAABB[t * 2 + 1]; }
    else       { lmn = bvhData[li * 2]; lmx = bvhData[li * 2 + 1]; }
    if (rLeaf) { const uint32_t t = sortedIdx[ri]; rmn = triAABB[t * 2]; rmx = triAABB[t * 2 + 1]; }
    else       { rmn = bvhData[ri * 2]; rmx = bvhData[ri * 2 + 1]; }
    bvhData[i * 2 + 0] = make_float4(rt_min(lmn.x, rmn.x), rt_min(lmn.y, rmn.y), rt_min(lmn.z, rmn.z), 0.0f);
    bvhData[i * 2 + 1] = make_float4(rt_max(lmx.x, rmx.x), rt_max(lmx.y, rmx.y), rt_max(lmx.z, rmx.z), 0.0f);
    if (nodes) {               // the exact 64-byte node: both child boxes, the links, the parent (for the stackless exact walk)
        nodes[i * 4 + 0] = make_float4(lmn.x, lmn.y, lmn.z, lmx.x);
        nodes[i * 4 + 1] = make_float4(lmx.y, lmx.z, rmn.x, rmn.y);
        nodes[i * 4 + 2] = make_float4(rmn.z, rmx.x, rmx.y, rmx.z);
        nodes[i * 4 + 3] = make_float4(__uint_as_float(l), __uint_as_float(r), __uint_as_float(parent[i]), 0.0f);
    }
    done[i] = iter;
    }
}

// A level-synchronous pass looks at every node in every launch although half of the nodes are done after the
// first launch and seven eighths after the third.  From then on the launches run over a list of the nodes
// still to do, renewed every few levels (the order of the list does not matter).
#define COMPACT_THREADS 1024
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact_undone(uint32_t nAll, const uint32_t *__restrict__ listIn, const uint32_t *__restrict__ countIn,
                                                                    const uint32_t *__restrict__ done, uint32_t *__restrict__ listOut,
                                                                    uint32_t *__restrict__ count)
{
    __shared__ uint32_t waveBase[COMPACT_THREADS / 64];
    __shared__ uint32_t blockBase;
    const uint32_t n = listIn ? *countIn : nAll;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x * COMPACT_THREADS; chunk < n; chunk += gridDim.x * COMPACT_THREADS) {
        const uint32_t k = chunk + threadIdx.x;
        const uint32_t i = k < n ? (listIn ? listIn[k] : k) : 0u;
        const bool keep = k < n && done[i] == FIT_UNDONE;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) waveBase[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {          // one atomic per chunk: atomics on one word are served one at a time
            uint32_t total = 0;
            for (uint32_t w = 0; w < COMPACT_THREADS / 64; ++w) {
                const uint32_t c = waveBase[w];
                waveBase[w] = total;
                total += c;
            }
            blockBase = total ? atomicAdd(count, total) : 0u;
        }
        __syncthreads();
        if (keep) listOut[blockBase + waveBase[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ traversal tree: fit and tree rotations in one sweep
// The tree the rays walk gets its boxes (pass 1, the union k_fit computes) and then RT_ROTATION_PASSES bottom-up
// passes of include/rt_rotate.h (passes 2, 3, ...).  Run pass after pass and level after level that is
// (1 + passes) x height launches.  The passes overlap instead, as a wavefront: in every launch of the sweep a
// node takes its NEXT pass p if, in EARLIER launches,
//   (a) its internal children finished pass p (their subtrees are as pass p leaves them), and
//   (b) for p >= 3, its parent finished pass p - 1 (the only other node whose rotation rewrites this node's
//       links and box has made that pass's change to them).
// A rotation at node i reads the links of i and of its children and the boxes of its children and
// grandchildren, and rewrites the links of i, the links and box of the child that changes and the parent words
// of what moved.  With (a) and (b) nothing it reads can be rewritten in the same launch (i's parent waits for i
// by (a), i's children wait for i by (b), a grandparent's rotation moves i as a whole, and a node's own
// rotation never changes its own box), every read of another workgroup's result is ordered by a kernel
// boundary, and every decision sees what it sees when the passes run one after the other: the outcome is the
// one of the serial, post-order passes of the checker, in height + 2 x passes launches.
// prog[i] = passes finished << 8 | launch that finished the last one (launches count from 1).
#define SWEEP_PASSES(w) ((w) >> 8)
#define SWEEP_LAUNCH(w) ((w) & 255u)
#define SWEEP_MAX_LAUNCH 255u
__device__ __forceinline__ bool sweep_finished_before(uint32_t w, uint32_t pass, uint32_t iter)
{
    return SWEEP_PASSES(w) > pass || (SWEEP_PASSES(w) == pass && SWEEP_LAUNCH(w) < iter);
}

// leafBox: the triangles' boxes in the order of this tree's leaves (k_gather_leaf_boxes), so that the leaves under a
// node, neighbours in that order, are neighbours in memory in every pass instead of scattered over the input order
__device__ __forceinline__ void load_box(uint32_t child, const float4 *__restrict__ leafBox, const float4 *bvhData, float out[6])
{
    float4 mn, mx;
    if (child & RT_LEAF_BIT) {
        const uint32_t t = child & ~RT_LEAF_BIT;
        mn = leafBox[t * 2];
        mx = leafBox[t * 2 + 1];
    } else {
        mn = bvhData[child * 2];
        mx = bvhData[child * 2 + 1];
    }
    out[0] = mn.x; out[1] = mn.y; out[2] = mn.z; out[3] = mx.x; out[4] = mx.y; out[5] = mx.z;
}

__global__ __launch_bounds__(256) void k_gather_leaf_boxes(uint32_t N, const uint32_t *__restrict__ sortedIdx, const float4 *__restrict__ triAABB,
                                                            float4 *__restrict__ leafBox)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t t = sortedIdx[i];
    leafBox[i * 2] = triAABB[t * 2];
    leafBox[i * 2 + 1] = triAABB[t * 2 + 1];
}

__device__ __forceinline__ void store_box(float4 *bvhData, uint32_t node, const float m[6])
{
    bvhData[node * 2 + 0] = make_float4(m[0], m[1], m[2], 0.0f);
    bvhData[node * 2 + 1] = make_float4(m[3], m[4], m[5], 0.0f);
}

__device__ __forceinline__ void set_parent(uint32_t *parent, uint32_t child, uint32_t to)
{
    if (!(child & RT_LEAF_BIT)) parent[child] = to;
}

__device__ __forceinline__ void rotate_node(uint32_t i, uint32_t l, uint32_t r, uint32_t *left, uint32_t *right, uint32_t *parent,
                                            const float4 *__restrict__ leafBox, float4 *bvhData)
{
    const bool lInt = !(l & RT_LEAF_BIT), rInt = !(r & RT_LEAF_BIT);
    float L[6], R[6], La[6] = {0, 0, 0, 0, 0, 0}, Lb[6] = {0, 0, 0, 0, 0, 0}, Ra[6] = {0, 0, 0, 0, 0, 0}, Rb[6] = {0, 0, 0, 0, 0, 0};
    uint32_t la = 0, lb = 0, ra = 0, rb = 0;
    load_box(l, leafBox, bvhData, L);
    load_box(r, leafBox, bvhData, R);
    if (lInt) {
        la = left[l]; lb = right[l];
        load_box(la, leafBox, bvhData, La);
        load_box(lb, leafBox, bvhData, Lb);
    }
    if (rInt) {
        ra = left[r]; rb = right[r];
        load_box(ra, leafBox, bvhData, Ra);
        load_box(rb, leafBox, bvhData, Rb);
    }
    const int which = rt_pick_rotation(lInt, rInt, L, R, La, Lb, Ra, Rb);
    float m[6], n2[6];
    switch (which) {
    case 1:                      // L <-> R.left: R = (L, Rb)
        left[i] = ra; left[r] = l;
        set_parent(parent, ra, i); set_parent(parent, l, r);
        rt_box_merge(L, Rb, m); store_box(bvhData, r, m);
        break;
    case 2:                      // L <-> R.right: R = (Ra, L)
        left[i] = rb; right[r] = l;
        set_parent(parent, rb, i); set_parent(parent, l, r);
        rt_box_merge(Ra, L, m); store_box(bvhData, r, m);
        break;
    case 3:                      // R <-> L.left: L = (R, Lb)
        right[i] = la; left[l] = r;
        set_parent(parent, la, i); set_parent(parent, r, l);
        rt_box_merge(R, Lb, m); store_box(bvhData, l, m);
        break;
    case 4:                      // R <-> L.right: L = (La, R)
        right[i] = lb; right[l] = r;
        set_parent(parent, lb, i); set_parent(parent, r, l);
        rt_box_merge(La, R, m); store_box(bvhData, l, m);
        break;
    case 5:                      // La <-> Ra: both children change
        left[l] = ra; left[r] = la;
        set_parent(parent, ra, l); set_parent(parent, la, r);
        rt_box_merge(Ra, Lb, m); rt_box_merge(La, Rb, n2);
        store_box(bvhData, l, m); store_box(bvhData, r, n2);
        break;
    case 6:                      // La <-> Rb
        left[l] = rb; right[r] = la;
        set_parent(parent, rb, l); set_parent(parent, la, r);
        rt_box_merge(Rb, Lb, m); rt_box_merge(Ra, La, n2);
        store_box(bvhData, l, m); store_box(bvhData, r, n2);
        break;
    default: break;
    }
}

__global__ __launch_bounds__(256) void k_sweep(uint32_t nAll, const uint32_t *__restrict__ list, const uint32_t *__restrict__ listCount, uint32_t iter,
                                                uint32_t nPasses, uint32_t *left, uint32_t *right, uint32_t *parent,
                                                const float4 *__restrict__ leafBox, float4 *bvhData, uint32_t *prog)
{
    const uint32_t n = list ? *listCount : nAll;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t i = list ? list[k] : k;
        const uint32_t mine = prog[i];
        if (SWEEP_PASSES(mine) >= nPasses) continue;
        const uint32_t pass = SWEEP_PASSES(mine) + 1u;
        const uint32_t l = left[i], r = right[i];
        if (!(l & RT_LEAF_BIT) && !sweep_finished_before(prog[l], pass, iter)) continue;
        if (!(r & RT_LEAF_BIT) && !sweep_finished_before(prog[r], pass, iter)) continue;
        if (pass >= 3u) {
            const uint32_t up = parent[i];
            if (up != 0xFFFFFFFFu && !sweep_finished_before(prog[up], pass - 1u, iter)) continue;
        }
        if (pass == 1u) {
            float a[6], b[6], m[6];
            load_box(l, leafBox, bvhData, a);
            load_box(r, leafBox, bvhData, b);
            rt_box_merge(a, b, m);
            store_box(bvhData, i, m);
        } else {
            rotate_node(i, l, r, left, right, parent, leafBox, bvhData);
        }
        prog[i] = pass << 8 | iter;
    }
}

// The nodes of a sweep that still have passes to take (the order of the list does not matter).
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact_sweep(uint32_t nAll, const uint32_t *__restrict__ listIn, const uint32_t *__restrict__ countIn,
                                                                   const uint32_t *__restrict__ prog, uint32_t nPasses, uint32_t *__restrict__ listOut,
                                                                   uint32_t *__restrict__ count)
{
    __shared__ uint32_t waveBase[COMPACT_THREADS / 64];
    __shared__ uint32_t blockBase;
    const uint32_t n = listIn ? *countIn : nAll;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x * COMPACT_THREADS; chunk < n; chunk += gridDim.x * COMPACT_THREADS) {
        const uint32_t k = chunk + threadIdx.x;
        const uint32_t i = k < n ? (listIn ? listIn[k] : k) : 0u;
        const bool keep = k < n && SWEEP_PASSES(prog[i]) < nPasses;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) waveBase[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < COMPACT_THREADS / 64; ++w) {
                const uint32_t c = waveBase[w];
                waveBase[w] = total;
                total += c;
            }
            blockBase = total ? atomicAdd(count, total) : 0u;
        }
        __syncthreads();
        if (keep) listOut[blockBase + waveBase[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}

// What the second half of the build derives from the scene's box (the root box of the reference's tree): made on
// the device by one thread, so the host does not have to wait for that box in the middle of the build.
struct BuildPlan {
    RtKeyPlan keys;
    RtQGrid grid;
    RtAABB root;
    uint32_t smallEdgeProduct, smallEdgeSum;     // float bits: the largest among the small triangles (k_pack_leaves)
};

__global__ void k_plan(const float4 *__restrict__ bvhData, BuildPlan *__restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float4 mn = bvhData[0], mx = bvhData[1];
    const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
    BuildPlan p;
    p.keys = rt_key_plan(lo, hi);
    p.grid = rt_qh_grid(lo, hi);
    memset(&p.root, 0, sizeof p.root);
    for (int a = 0; a < 3; ++a) {
        p.root.min[a] = lo[a];
        p.root.max[a] = hi[a];
    }
    p.smallEdgeProduct = 0u;
    p.smallEdgeSum = 0u;
    *out = p;
}

// ------------------------------------------------------------------ half-float traversal nodes
// The 32-byte node of a tree whose links and boxes are final: both child boxes rounded outward onto the
// half-float planes of the root-box grid (rt_qh.h).  A plane the grid cannot cover raises the error flag.
__global__ __launch_bounds__(256) void k_emit_qnodes(uint32_t N, const BuildPlan *__restrict__ planPtr, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                      const float4 *__restrict__ leafBox,
                                                      const float4 *__restrict__ bvhData, uint4 *__restrict__ qnodes, uint32_t *__restrict__ errorFlag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N - 1) return;
    const RtQGrid g = planPtr->grid;
    const uint32_t l = left[i], r = right[i];
    float a[6], b[6];
    load_box(l, leafBox, bvhData, a);
    load_box(r, leafBox, bvhData, b);
    uint32_t wl[3], wr[3];
    const bool okL = rt_qh_box(g, a, a + 3, wl);
    const bool okR = rt_qh_box(g, b, b + 3, wr);
    if (!okL || !okR) *errorFlag = 2u;
    qnodes[i * 2 + 0] = make_uint4(wl[0], wl[1], wl[2], wr[0]);
    qnodes[i * 2 + 1] = make_uint4(wr[1], wr[2], l, r);
}

// Two levels per record: the places of node i hold the children of its left child, then of its right child; a child
// that is a leaf takes one place as itself.  Boxes and references are copied from the finished 32-byte nodes.
// large: per leaf / per node, "is (holds) a large triangle" -- carried in bit 30 of the references.
__global__ __launch_bounds__(256) void k_emit_wide_nodes(uint32_t NI, const uint4 *__restrict__ qnodes, uint4 *__restrict__ wnodes,
                                                          const uint32_t *__restrict__ leafLarge, const uint32_t *__restrict__ nodeLarge)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NI) return;
    const uint4 a = qnodes[i * 2], b = qnodes[i * 2 + 1];
    const uint32_t cbox[2][3] = {{a.x, a.y, a.z}, {a.w, b.x, b.y}}, cref[2] = {b.z, b.w};
    uint32_t box[4][3], ref[4];
    for (int k = 0; k < 4; ++k) {
        box[k][0] = box[k][1] = box[k][2] = RT_QH_NO_BOX;
        ref[k] = RT_INVALID;
    }
    int n = 0;
    for (int c = 0; c < 2; ++c) {
        if (cref[c] & RT_LEAF_BIT) {
            for (int x = 0; x < 3; ++x) box[n][x] = cbox[c][x];
            ref[n++] = cref[c];
        } else {
            const uint4 ga = qnodes[cref[c] * 2], gb = qnodes[cref[c] * 2 + 1];
            box[n][0] = ga.x; box[n][1] = ga.y; box[n][2] = ga.z;
            ref[n++] = gb.z;
            box[n][0] = ga.w; box[n][1] = gb.x; box[n][2] = gb.y;
            ref[n++] = gb.w;
        }
    }
    for (int k = 0; k < 4; ++k)
        if (ref[k] != RT_INVALID && ((ref[k] & RT_LEAF_BIT) ? leafLarge[ref[k] & ~RT_LEAF_BIT] : nodeLarge[ref[k]]) != 0u) ref[k] |= RT_REF_LARGE;
    wnodes[i * 4 + 0] = make_uint4(box[0][0], box[0][1], box[0][2], box[1][0]);
    wnodes[i * 4 + 1] = make_uint4(box[1][1], box[1][2], box[2][0], box[2][1]);
    wnodes[i * 4 + 2] = make_uint4(box[2][2], box[3][0], box[3][1], box[3][2]);
    wnodes[i * 4 + 3] = make_uint4(ref[0], ref[1], ref[2], ref[3]);
}

// ------------------------------------------------------------------ leaf packing
__global__ __launch_bounds__(256) void k_pack_leaves(uint32_t N, const uint32_t *__restrict__ sortedIdx,
                                                      const uint32_t *__restrict__ idx, const float *__restrict__ verts,
                                                      const uint32_t *__restrict__ matIdx, const int32_t *__restrict__ shadow,
                                                      const float4 *__restrict__ triAABB, float4 *__restrict__ leafTri,
                                                      float4 *__restrict__ leafShade, uint32_t *__restrict__ leafLarge, BuildPlan *plan)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    // (leafLarge, plan: the leaf's class for the bounded rule and the two maxima over the small ones; nullptr: not wanted)
    float product = 0.0f, sum = 0.0f;
    if (j < N) {
    const uint32_t t = sortedIdx[j];
    const float4 bmn = triAABB[t * 2], bmx = triAABB[t * 2 + 1];
    const uint32_t i0 = idx[t * 3 + 0], i1 = idx[t * 3 + 1], i2 = idx[t * 3 + 2];
    const float *a = verts + (size_t)i0 * 11, *b = verts + (size_t)i1 * 11, *c = verts + (size_t)i2 * 11;
    const rt_f3 v0 = rt_mk3(a[0], a[1], a[2]);
    const rt_f3 e1 = rt_sub3(rt_mk3(b[0], b[1], b[2]), v0);
    const rt_f3 e2 = rt_sub3(rt_mk3(c[0], c[1], c[2]), v0);
    leafTri[j * 4 + 0] = make_float4(v0.x, v0.y, v0.z, e1.x);
    leafTri[j * 4 + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    leafTri[j * 4 + 2] = make_float4(e2.z, __uint_as_float(t), bmn.x, bmn.y);
    leafTri[j * 4 + 3] = make_float4(bmn.z, bmx.x, bmx.y, bmx.z);
    leafShade[j * 3 + 0] = make_float4(a[3], a[4], a[5], b[3]);
    leafShade[j * 3 + 1] = make_float4(b[4], b[5], c[3], c[4]);
    leafShade[j * 3 + 2] = make_float4(c[5], __uint_as_float(matIdx[t]), __int_as_float(shadow[t * 2 + 0]), 0.0f);
    if (leafLarge) {
        // rounded up generously (2^-20 relative): the bound they go into has to hold for the exact lengths
        const float l1 = rt_sqrt(rt_dot3(e1, e1)) * 1.000001f, l2 = rt_sqrt(rt_dot3(e2, e2)) * 1.000001f;
        product = l1 * l2 * 1.000001f;
        sum = (l1 + l2) * 1.000001f;
        const bool small = product <= RT_SMALL_EDGE_PRODUCT && sum <= RT_SMALL_EDGE_SUM;     // (false for NaN)
        leafLarge[j] = small ? 0u : 1u;
        if (!small) product = sum = 0.0f;
    }
    }
    if (plan) {        // one atomic pair per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            product = rt_max(product, __shfl_down(product, off, 64));
            sum = rt_max(sum, __shfl_down(sum, off, 64));
        }
        // (a look first: the maxima only grow, and atomics on one word are served one at a time -- a pair from each of
        //  the car's 7 750 waves took 0.16 ms)
        if ((threadIdx.x & 63u) == 0u) {
            const volatile uint32_t *seen = &plan->smallEdgeProduct;
            if (__float_as_uint(product) > seen[0]) atomicMax(&plan->smallEdgeProduct, __float_as_uint(product));
            if (__float_as_uint(sum) > seen[1]) atomicMax(&plan->smallEdgeSum, __float_as_uint(sum));
        }
    }
}

// Every node on the way from a large triangle to the root holds one: a thread per node looks at its two children and,
// where one is a large leaf, marks the node and its ancestors (plain stores of the same value from whoever gets there).
__global__ __launch_bounds__(256) void k_mark_large(uint32_t NI, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                     const uint32_t *__restrict__ parent, const uint32_t *__restrict__ leafLarge,
                                                     uint32_t *nodeLarge)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NI) return;
    const uint32_t l = left[i], r = right[i];
    const bool any = ((l & RT_LEAF_BIT) && leafLarge[l & ~RT_LEAF_BIT]) || ((r & RT_LEAF_BIT) && leafLarge[r & ~RT_LEAF_BIT]);
    if (!any) return;
    uint32_t at = i;
    // (a node found marked has a marker on its way to the root from there already)
    for (uint32_t guard = 0; guard < 4096u && at != 0xFFFFFFFFu; ++guard) {
        if (((const volatile uint32_t *)nodeLarge)[at] != 0u && at != i) break;
        nodeLarge[at] = 1u;
        at = parent[at];
    }
}

// ------------------------------------------------------------------ wire-struct readback kernels
__global__ void k_wire_nodes(uint32_t N, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                             const uint32_t *__restrict__ parent, const uint32_t *__restrict__ leafParent,
                             RtInternalNode *__restrict__ inodes, RtLeafNode *__restrict__ leaves)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        RtLeafNode ln;
        ln.parent = leafParent[i];
        ln.index = i;
        leaves[i] = ln;
    }
    if (i + 1 < N) {
        RtInternalNode n;
        const uint32_t l = left[i], r = right[i];
        n.leftNode = l & ~RT_LEAF_BIT;
        n.leftNodeType = (l & RT_LEAF_BIT) ? RT_LEAF_NODE : RT_INTERNAL_NODE;
        n.rightNode = r & ~RT_LEAF_BIT;
        n.rightNodeType = (r & RT_LEAF_BIT) ? RT_LEAF_NODE : RT_INTERNAL_NODE;
        n.parent = parent[i];
        n.index = i;
        inodes[i] = n;
    }
}

// MeshData.compute:108-130: the 192-byte Triangle record, produced on demand.
__global__ void k_wire_triangles(uint32_t N, const uint32_t *__restrict__ idx, const float *__restrict__ verts,
                                 const uint32_t *__restrict__ matIdx, const int32_t *__restrict__ shadow,
                                 RtTriangle *__restrict__ out)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= N) return;
    RtTriangle t;
    memset(&t, 0, sizeof t);
    const float *v[3];
    for (int k = 0; k < 3; ++k) v[k] = verts + (size_t)idx[id * 3 + k] * 11;
    for (int c = 0; c < 3; ++c) {
        t.point0[c] = v[0][c]; t.point1[c] = v[1][c]; t.point2[c] = v[2][c];
        t.normal0[c] = v[0][3 + c]; t.normal1[c] = v[1][3 + c]; t.normal2[c] = v[2][3 + c];
        t.tangent0[c] = v[0][6 + c]; t.tangent1[c] = v[1][6 + c]; t.tangent2[c] = v[2][6 + c];
    }
    for (int c = 0; c < 2; ++c) { t.uv0[c] = v[0][9 + c]; t.uv1[c] = v[1][9 + c]; t.uv2[c] = v[2][9 + c]; }
    t.materialIndex = (int32_t)matIdx[id];
    t.castShadow = shadow[id * 2 + 0];
    t.receiveShadow = shadow[id * 2 + 0];      // MeshData.compute:127-128 reads [0] twice
    out[id] = t;
}

// ------------------------------------------------------------------ the traversal tree's own keys and bookkeeping
// Sort key of the tree the rays walk (include/rt_finekey.h): the centre of the triangle's box,
// the same (min + max) * 0.5 as k_calculate, on a 29-bit grid over the scene's own box, under 2 bits for the size of the box.
__global__ __launch_bounds__(256) void k_fine_keys(uint32_t N, const float4 *__restrict__ triAABB, const BuildPlan *__restrict__ planPtr,
                                                    uint32_t *__restrict__ keys, uint32_t *__restrict__ triIndex)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= N) return;
    const RtKeyPlan plan = planPtr->keys;
    const float4 mn = triAABB[id * 2], mx = triAABB[id * 2 + 1];
    const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
    keys[id] = rt_fine_key(plan, lo, hi);
    triIndex[id] = id;
}

// Position of every triangle in the reference's visiting order (BVHRayTracing.cginc:78-123: at a
// node the left child is handled first if it is a leaf, then the right one if it is a leaf, then
// the right subtree, then the left subtree), written at the triangle's leaf in the traversal tree.
__global__ __launch_bounds__(256) void k_visit_rank(uint32_t N, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                     const uint32_t *__restrict__ parent, const uint32_t *__restrict__ leafParent,
                                                     const uint32_t *__restrict__ size, const uint32_t *__restrict__ refSortedIdx,
                                                     const uint32_t *__restrict__ walkLeafOfTri, uint32_t *__restrict__ leafRank)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    uint32_t rank = 0, child = j | RT_LEAF_BIT, x = leafParent[j];
    while (x != RT_INVALID) {
        const uint32_t l = left[x], r = right[x];
        const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (r & RT_LEAF_BIT) != 0;
        if (l == child) rank += lLeaf ? 0u : (rLeaf ? 1u : size[r]);
        else rank += lLeaf ? 1u : 0u;
        child = x;
        x = parent[x];
    }
    leafRank[walkLeafOfTri[refSortedIdx[j]]] = rank;
}

__global__ __launch_bounds__(256) void k_invert_leaves(uint32_t N, const uint32_t *__restrict__ walkSortedIdx, uint32_t *__restrict__ walkLeafOfTri)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < N) walkLeafOfTri[walkSortedIdx[l]] = l;
}

// The exact nodes of the reference tree (walked by rays with a zero / non-finite direction component)
// name their leaf children by the traversal tree's leaf numbers, where the triangle records live.
__global__ __launch_bounds__(256) void k_remap_leaf_children(uint32_t NI, const uint32_t *__restrict__ refSortedIdx,
                                                              const uint32_t *__restrict__ walkLeafOfTri, float4 *__restrict__ nodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NI) return;
    float4 w = nodes[i * 4 + 3];
    uint32_t l = __float_as_uint(w.x), r = __float_as_uint(w.y);
    if (l & RT_LEAF_BIT) l = walkLeafOfTri[refSortedIdx[l & ~RT_LEAF_BIT]] | RT_LEAF_BIT;
    if (r & RT_LEAF_BIT) r = walkLeafOfTri[refSortedIdx[r & ~RT_LEAF_BIT]] | RT_LEAF_BIT;
    w.x = __uint_as_float(l);
    w.y = __uint_as_float(r);
    nodes[i * 4 + 3] = w;
}

// ------------------------------------------------------------------ host orchestration
void rt_free_build(RtContext *ctx)
{
    void **ptrs[] = {(void **)&ctx->dTriAABB, (void **)&ctx->dMortonUnsorted, (void **)&ctx->dMortonSortedRaw,
                     (void **)&ctx->dMorton, (void **)&ctx->dSortedIdx, (void **)&ctx->dLeft, (void **)&ctx->dRight,
                     (void **)&ctx->dParent, (void **)&ctx->dLeafParent, (void **)&ctx->dBvhData, (void **)&ctx->dNodes, (void **)&ctx->dQNodes, (void **)&ctx->dWNodes,
                     (void **)&ctx->dLeafTri, (void **)&ctx->dLeafShade, (void **)&ctx->dLeafRank};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (auto &b : ctx->buildScratch)
        if (b.first) (void)hipFree(b.first);
    ctx->buildScratch.clear();
    ctx->buildBuffersN = 0;
    ctx->built = false;
}

// A build's scratch: asked for in the same order with the same sizes by every build of the same triangle count, so the
// k-th request of a build takes the k-th buffer the last one left if its size is the same.  What a build leaves is kept
// while it is no more than RT_BUILD_SCRATCH_KEEP bytes; more than that is freed when the build ends, however it ends.
struct RtBuildScratchPool {
    RtContext *ctx;
    size_t next = 0;
    explicit RtBuildScratchPool(RtContext *c) : ctx(c) {}
    RtBuildScratchPool(const RtBuildScratchPool &) = delete;
    RtBuildScratchPool &operator=(const RtBuildScratchPool &) = delete;
    template <typename T>
    hipError_t alloc(T **p, size_t count)
    {
        const size_t bytes = (count ? count : 1) * sizeof(T);
        auto &pool = ctx->buildScratch;
        if (next == pool.size()) pool.emplace_back(nullptr, 0);
        auto &slot = pool[next];
        if (slot.first == nullptr || slot.second != bytes) {
            if (slot.first) (void)hipFree(slot.first);
            slot.first = nullptr;
            slot.second = 0;
            const hipError_t e = hipMalloc(&slot.first, bytes);
            if (e != hipSuccess) return e;
            slot.second = bytes;
        }
        *p = (T *)slot.first;
        next += 1;
        return hipSuccess;
    }
    ~RtBuildScratchPool()
    {
        size_t total = 0;
        for (const auto &b : ctx->buildScratch) total += b.second;
        if (total <= RT_BUILD_SCRATCH_KEEP) return;
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &b : ctx->buildScratch)
            if (b.first) (void)hipFree(b.first);
        ctx->buildScratch.clear();
    }
};

template <typename T>
static hipError_t dalloc(T **p, size_t count)
{
    return hipMalloc((void **)p, (count ? count : 1) * sizeof(T));
}

// Scratch shared by the two passes of the pipeline.
struct BuildScratch {
    uint32_t *keysB = nullptr, *valsB = nullptr, *blockHist = nullptr, *scanSums = nullptr, *done = nullptr;
    uint32_t *undone[2] = {nullptr, nullptr}, *counts = nullptr;   // counts: PASS_COUNT_WORDS words, one per compaction of the build
    uint32_t nextCount = 0, finalCount[32], nFinal = 0;           // the words that must read 0 when the build has finished
    uint32_t nbSort = 0, histLen = 0;
};

// The bottom-up fit of the reference's tree, a launch per tree level: launch(blocks, list, listCount, iter).
// From the third level on the launches run over a list of the nodes still to do, renewed every six levels; the
// length of a list stays on the device (every compaction has a count word of its own in sc.counts, zeroed once per
// build), the launches over it are grid-stride loops of a size the host only estimates, so no level waits for the
// host.  FIT_LEVELS levels are launched whatever the tree's height: along a root-to-leaf chain of a Karras tree
// the common prefix of the unique 32-bit keys grows by at least a bit per node, so no chain has more than 32
// internal nodes.  The sweep of the other tree (run_sweep) gets PASS_LEVELS, room for the levels its rotations
// may add; the count of what either leaves undone is looked at when the build ends.
#define FIT_LEVELS 34u
#define PASS_LEVELS 50u
#define PASS_COUNT_WORDS 256u
template <class Launch>
static int run_levels(RtContext *ctx, BuildScratch &sc, uint32_t NI, Launch launch)
{
    hipStream_t st = ctx->stream;
    RT_HIP(ctx, hipMemsetAsync(sc.done, 0xFF, (size_t)NI * 4, st));
    const uint32_t *list = nullptr, *listCount = nullptr;
    uint32_t expected = NI;       // a guess at the list's length, for the size of the launches only
    int which = 0;
    for (uint32_t iter = 0; iter < FIT_LEVELS; ++iter) {
        const uint32_t blocks = list ? (expected + 255u) / 256u : (NI + 255u) / 256u;
        launch(blocks < 64u ? 64u : blocks, list, listCount, iter);
        if (iter == 2 || (iter > 2 && (iter - 2) % 6 == 0)) {
            if (sc.nextCount >= PASS_COUNT_WORDS) return rt_fail(ctx, RT_E_STATE, "BVH build: out of count words");
            uint32_t *count = sc.counts + sc.nextCount++;
            const uint32_t cblocks = ((list ? expected : NI) + COMPACT_THREADS - 1) / COMPACT_THREADS;
            hipLaunchKernelGGL(k_compact_undone, dim3(cblocks < 16u ? 16u : cblocks), dim3(COMPACT_THREADS), 0, st, NI, list, listCount, sc.done,
                               sc.undone[which], count);
            list = sc.undone[which];
            listCount = count;
            which ^= 1;
            expected = iter == 2 ? NI / 4u : expected / 16u;
            if (expected < 4096u) expected = 4096u;
        }
    }
    // what is undone after the last level must be nothing: counted now, looked at by rt_build_impl after its final synchronisation
    if (sc.nFinal >= 32 || sc.nextCount >= PASS_COUNT_WORDS) return rt_fail(ctx, RT_E_STATE, "BVH build: too many passes");
    uint32_t *count = sc.counts + sc.nextCount++;
    hipLaunchKernelGGL(k_compact_undone, dim3(16), dim3(COMPACT_THREADS), 0, st, NI, list, listCount, sc.done, sc.undone[which], count);
    sc.finalCount[sc.nFinal++] = (uint32_t)(count - sc.counts);
    return RT_OK;
}

// The sweep of k_sweep over the NI internal nodes: nPasses passes, launches 1 .. PASS_LEVELS + what the passes add.
// The nodes at the bottom finish their last pass in launch `firstFinish`, and from there on half of what is left
// finishes with every launch: the launches then run over a list of the unfinished nodes, renewed after every launch
// at first and less often as it shrinks.  As in run_levels the list lengths stay on the device.
static int run_sweep(RtContext *ctx, BuildScratch &sc, uint32_t NI, uint32_t nPasses, uint32_t *left, uint32_t *right, uint32_t *parent,
                     const float4 *leafBox, RtAABB *bvhData)
{
    hipStream_t st = ctx->stream;
    RT_HIP(ctx, hipMemsetAsync(sc.done, 0, (size_t)NI * 4, st));
    const uint32_t firstFinish = nPasses >= 3u ? 2u * nPasses - 2u : nPasses;
    const uint32_t last = PASS_LEVELS + firstFinish;
    if (last > SWEEP_MAX_LAUNCH) return rt_fail(ctx, RT_E_STATE, "BVH build: too many passes for one sweep");
    const uint32_t *list = nullptr, *listCount = nullptr;
    uint32_t expected = NI, nextCompact = firstFinish, compactions = 0;
    int which = 0;
    for (uint32_t iter = 1; iter <= last; ++iter) {
        const uint32_t blocks = ((list ? expected : NI) + 255u) / 256u;
        rt_timer_begin(ctx, "k_sweep");
        hipLaunchKernelGGL(k_sweep, dim3(blocks < 64u ? 64u : blocks), dim3(256), 0, st, NI, list, listCount, iter, nPasses, left, right, parent, leafBox,
                           (float4 *)bvhData, sc.done);
        rt_timer_end(ctx);
        if (iter == nextCompact && iter < last) {
            if (sc.nextCount >= PASS_COUNT_WORDS) return rt_fail(ctx, RT_E_STATE, "BVH build: out of count words");
            uint32_t *count = sc.counts + sc.nextCount++;
            const uint32_t cblocks = ((list ? expected : NI) + COMPACT_THREADS - 1) / COMPACT_THREADS;
            hipLaunchKernelGGL(k_compact_sweep, dim3(cblocks < 16u ? 16u : cblocks), dim3(COMPACT_THREADS), 0, st, NI, list, listCount, sc.done, nPasses,
                               sc.undone[which], count);
            list = sc.undone[which];
            listCount = count;
            which ^= 1;
            compactions += 1;
            // a guess from above (blocks that find nothing to do cost next to nothing, too few blocks cost a lot): a balanced
            // tree would halve the list with every launch, a Karras tree has more tall nodes than that
            const uint32_t shrink = (iter - firstFinish + 1u) * 2u / 3u;
            expected = shrink < 20u ? NI >> shrink : 0u;
            if (expected < 65536u) expected = 65536u;
            nextCompact += compactions < 5u ? 1u : (compactions < 9u ? 2u : 6u);
        }
    }
    if (sc.nFinal >= 32 || sc.nextCount >= PASS_COUNT_WORDS) return rt_fail(ctx, RT_E_STATE, "BVH build: too many passes");
    uint32_t *count = sc.counts + sc.nextCount++;
    hipLaunchKernelGGL(k_compact_sweep, dim3(16), dim3(COMPACT_THREADS), 0, st, NI, list, listCount, sc.done, nPasses, sc.undone[which], count);
    sc.finalCount[sc.nFinal++] = (uint32_t)(count - sc.counts);
    return RT_OK;
}

// sort -> duplicate-key spreading -> Karras tree -> bottom-up fit, on keys already in keysRaw / ids in sortedIdx.
// Outputs: keysRaw (sorted), sortedIdx, codes (unique), left/right/parent/leafParent, bvhData, nodes (exact 64-byte nodes).
static int build_tree_from_keys(RtContext *ctx, BuildScratch &sc, uint32_t N, uint32_t *keysRaw, uint32_t *sortedIdx, uint32_t *codes,
                                uint32_t *left, uint32_t *right, uint32_t *parent, uint32_t *leafParent, uint32_t *rangeSize,
                                RtAABB *bvhData, RtNode64 *nodes, bool fit = true)
{
    hipStream_t st = ctx->stream;
    const uint32_t NI = N - 1;
    const dim3 blk(256), gridNI((NI + 255) / 256);
    uint32_t *keysA = keysRaw, *valsA = sortedIdx, *keysB = sc.keysB, *valsB = sc.valsB;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        rt_timer_begin(ctx, "k_sort_hist");
        hipLaunchKernelGGL(k_sort_hist, dim3(sc.nbSort), dim3(SORT_THREADS), 0, st, keysA, N, shift, sc.blockHist, sc.nbSort);
        rt_timer_end(ctx);
        launch_scan<0>(ctx, sc.blockHist, sc.blockHist, sc.histLen, sc.scanSums);
        rt_timer_begin(ctx, "k_sort_scatter");
        hipLaunchKernelGGL(k_sort_scatter, dim3(sc.nbSort), dim3(SORT_THREADS), 0, st, keysA, valsA, keysB, valsB, N, shift,
                           sc.blockHist, sc.nbSort);
        rt_timer_end(ctx);
        uint32_t *t = keysA; keysA = keysB; keysB = t;
        t = valsA; valsA = valsB; valsB = t;
    }
    // 4 passes: the result is back in keysRaw / sortedIdx.
    launch_scan<1>(ctx, keysRaw, codes, N, sc.scanSums);
    if (N < 2) {
        RT_HIP(ctx, hipMemsetAsync(leafParent, 0xFF, (size_t)N * 4, st));
        return RT_OK;
    }
    RT_HIP(ctx, hipMemsetAsync(parent, 0xFF, (size_t)NI * 4, st));
    RT_HIP(ctx, hipMemsetAsync(leafParent, 0xFF, (size_t)N * 4, st));
    rt_timer_begin(ctx, "k_tree");
    hipLaunchKernelGGL(k_tree, gridNI, blk, 0, st, codes, N, left, right, parent, leafParent, rangeSize);
    rt_timer_end(ctx);
    if (!fit) return RT_OK;
    return run_levels(ctx, sc, NI, [&](uint32_t blocks, const uint32_t *list, const uint32_t *listCount, uint32_t iter) {
        rt_timer_begin(ctx, "k_fit");
        hipLaunchKernelGGL(k_fit, dim3(blocks), blk, 0, st, NI, list, listCount, iter, left, right, sortedIdx, (const float4 *)ctx->dTriAABB,
                           (float4 *)bvhData, sc.done, (float4 *)nodes, parent);
        rt_timer_end(ctx);
    });
}

int rt_build_impl(RtContext *ctx)
{
    const uint32_t N = ctx->N;
    const bool reuse = N != 0 && ctx->buildBuffersN == N;       // the last build's buffers fit: as many triangles
    if (!reuse) rt_free_build(ctx);
    ctx->built = false;
    memset(&ctx->rootBox, 0, sizeof ctx->rootBox);
    ctx->smallEdgeProduct = ctx->smallEdgeSum = 0.0f;
    if (N >= RT_REF_INDEX) return rt_fail(ctx, RT_E_INVALID, "BVH build: at most 2^30 - 2 triangles");
    if (N == 0) {
        ctx->built = true;
        return RT_OK;
    }
    const uint32_t NI = N - 1;
    BuildScratch sc;
    sc.nbSort = (N + SORT_TILE - 1) / SORT_TILE;
    sc.histLen = 256u * sc.nbSort;
    const uint32_t nbScanMax = ((sc.histLen > N ? sc.histLen : N) + SCAN_TILE - 1) / SCAN_TILE;
    // the traversal tree's own build products, needed only until its nodes are quantised and its leaves packed
    uint32_t *wKeys = nullptr, *wCodes = nullptr, *wSortedIdx = nullptr, *wLeft = nullptr, *wRight = nullptr, *wParent = nullptr,
             *wLeafParent = nullptr, *refSize = nullptr, *walkLeafOfTri = nullptr;
    RtAABB *wBvhData = nullptr;
    BuildPlan *plan = nullptr;
    float4 *wLeafBox = nullptr;
    RtBuildScratchPool arena(ctx);

    RT_HIP(ctx, arena.alloc(&plan, 1));
    RT_HIP(ctx, arena.alloc(&wLeafBox, (size_t)N * 2));
    if (!reuse) {
        RT_HIP(ctx, dalloc(&ctx->dTriAABB, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dMortonUnsorted, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dMortonSortedRaw, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dMorton, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dSortedIdx, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dLeft, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dRight, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dParent, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dLeafParent, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dBvhData, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dNodes, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dQNodes, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dWNodes, (size_t)NI));
        RT_HIP(ctx, dalloc(&ctx->dLeafTri, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dLeafShade, (size_t)N));
        RT_HIP(ctx, dalloc(&ctx->dLeafRank, (size_t)N));
        ctx->buildBuffersN = N;
    }
    RT_HIP(ctx, arena.alloc(&sc.keysB, (size_t)N));
    RT_HIP(ctx, arena.alloc(&sc.valsB, (size_t)N));
    RT_HIP(ctx, arena.alloc(&sc.blockHist, (size_t)sc.histLen));
    RT_HIP(ctx, arena.alloc(&sc.scanSums, (size_t)nbScanMax + 1));
    RT_HIP(ctx, arena.alloc(&sc.done, (size_t)NI));
    RT_HIP(ctx, arena.alloc(&sc.undone[0], (size_t)NI));
    RT_HIP(ctx, arena.alloc(&sc.undone[1], (size_t)NI));
    RT_HIP(ctx, arena.alloc(&sc.counts, (size_t)PASS_COUNT_WORDS));
    RT_HIP(ctx, arena.alloc(&wKeys, (size_t)N));
    RT_HIP(ctx, arena.alloc(&wCodes, (size_t)N));
    RT_HIP(ctx, arena.alloc(&wSortedIdx, (size_t)N));
    RT_HIP(ctx, arena.alloc(&wLeft, (size_t)NI));
    RT_HIP(ctx, arena.alloc(&wRight, (size_t)NI));
    RT_HIP(ctx, arena.alloc(&wParent, (size_t)NI));
    RT_HIP(ctx, arena.alloc(&wLeafParent, (size_t)N));
    RT_HIP(ctx, arena.alloc(&refSize, (size_t)NI));
    RT_HIP(ctx, arena.alloc(&walkLeafOfTri, (size_t)N));
    RT_HIP(ctx, arena.alloc(&wBvhData, (size_t)NI));

    hipStream_t st = ctx->stream;
    for (hipEvent_t &e : ctx->buildEvents)
        if (!e) RT_HIP(ctx, hipEventCreate(&e));
    const hipEvent_t e0 = ctx->buildEvents[0], e1 = ctx->buildEvents[1];
    RT_HIP(ctx, hipEventRecord(e0, st));
    RT_HIP(ctx, hipMemsetAsync(sc.counts, 0, PASS_COUNT_WORDS * sizeof(uint32_t), st));

    const dim3 blk(256);
    const dim3 gridN((N + 255) / 256), gridNI((NI + 255) / 256);
    int rc;

    // ---- the reference's tree (what rt_readback_bvh returns; walked only by rays with a zero or
    //      non-finite direction component, and the source of every triangle's visiting rank)
    rt_timer_begin(ctx, "k_calculate");
    hipLaunchKernelGGL(k_calculate, gridN, blk, 0, st, ctx->dIdx, (const float *)ctx->dVerts, N, (float4 *)ctx->dTriAABB,
                       ctx->dMortonUnsorted, ctx->dSortedIdx);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipMemcpyAsync(ctx->dMortonSortedRaw, ctx->dMortonUnsorted, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    if ((rc = build_tree_from_keys(ctx, sc, N, ctx->dMortonSortedRaw, ctx->dSortedIdx, ctx->dMorton, ctx->dLeft, ctx->dRight, ctx->dParent,
                                   ctx->dLeafParent, refSize, ctx->dBvhData, ctx->dNodes)))
        return rc;

    if (N >= 2) {
        // ---- the tree the rays walk: same pipeline, keys on a fine grid over the scene's own box
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(64), 0, st, (const float4 *)ctx->dBvhData, plan);
        rt_timer_begin(ctx, "k_fine_keys");
        hipLaunchKernelGGL(k_fine_keys, gridN, blk, 0, st, N, (const float4 *)ctx->dTriAABB, plan, wKeys, wSortedIdx);
        rt_timer_end(ctx);
        if ((rc = build_tree_from_keys(ctx, sc, N, wKeys, wSortedIdx, wCodes, wLeft, wRight, wParent, wLeafParent, nullptr, wBvhData, nullptr, false)))
            return rc;
        // its boxes and the rotation passes (a rotation can lengthen a path by a level: PASS_LEVELS leaves room for that)
        hipLaunchKernelGGL(k_gather_leaf_boxes, gridN, blk, 0, st, N, wSortedIdx, (const float4 *)ctx->dTriAABB, wLeafBox);
        if ((rc = run_sweep(ctx, sc, NI, N >= 3 ? 1u + RT_ROTATION_PASSES : 1u, wLeft, wRight, wParent, wLeafBox, wBvhData))) return rc;
        // leaf records first: they say which triangles are large, and the two-level records carry that in their references
        rt_timer_begin(ctx, "k_pack_leaves");
        uint32_t *leafLarge = sc.keysB, *nodeLarge = sc.done;       // (scratch that the sorts and the sweep are done with)
        hipLaunchKernelGGL(k_pack_leaves, gridN, blk, 0, st, N, wSortedIdx, ctx->dIdx, (const float *)ctx->dVerts, ctx->dMatIdx, ctx->dShadow,
                           (const float4 *)ctx->dTriAABB, (float4 *)ctx->dLeafTri, (float4 *)ctx->dLeafShade, leafLarge, plan);
        rt_timer_end(ctx);
        RT_HIP(ctx, hipMemsetAsync(nodeLarge, 0, (size_t)NI * 4, st));
        rt_timer_begin(ctx, "k_emit_qnodes");
        hipLaunchKernelGGL(k_mark_large, gridNI, blk, 0, st, NI, wLeft, wRight, wParent, leafLarge, nodeLarge);
        hipLaunchKernelGGL(k_emit_qnodes, gridNI, blk, 0, st, N, plan, wLeft, wRight, wLeafBox,
                           (const float4 *)wBvhData, (uint4 *)ctx->dQNodes, ctx->dErrorFlag);
        hipLaunchKernelGGL(k_emit_wide_nodes, gridNI, blk, 0, st, NI, (const uint4 *)ctx->dQNodes, (uint4 *)ctx->dWNodes, leafLarge, nodeLarge);
        rt_timer_end(ctx);
        // ---- tie the two together: visiting rank per leaf, leaf numbers in the reference tree's exact nodes
        rt_timer_begin(ctx, "k_visit_rank");
        hipLaunchKernelGGL(k_invert_leaves, gridN, blk, 0, st, N, wSortedIdx, walkLeafOfTri);
        hipLaunchKernelGGL(k_visit_rank, gridN, blk, 0, st, N, ctx->dLeft, ctx->dRight, ctx->dParent, ctx->dLeafParent, refSize, ctx->dSortedIdx,
                           walkLeafOfTri, ctx->dLeafRank);
        hipLaunchKernelGGL(k_remap_leaf_children, gridNI, blk, 0, st, NI, ctx->dSortedIdx, walkLeafOfTri, (float4 *)ctx->dNodes);
        rt_timer_end(ctx);
    } else {
        // a single triangle: no internal node, nothing can be hit (bvhData[0] is the zero box); records in input order
        rt_timer_begin(ctx, "k_pack_leaves");
        hipLaunchKernelGGL(k_pack_leaves, gridN, blk, 0, st, N, ctx->dSortedIdx, ctx->dIdx, (const float *)ctx->dVerts, ctx->dMatIdx, ctx->dShadow,
                           (const float4 *)ctx->dTriAABB, (float4 *)ctx->dLeafTri, (float4 *)ctx->dLeafShade, nullptr, nullptr);
        rt_timer_end(ctx);
        RT_HIP(ctx, hipMemsetAsync(ctx->dLeafRank, 0, (size_t)N * 4, st));
    }

    RT_HIP(ctx, hipEventRecord(e1, st));
    uint32_t hostCounts[PASS_COUNT_WORDS];
    BuildPlan hostPlan;
    RT_HIP(ctx, hipMemcpyAsync(hostCounts, sc.counts, sizeof hostCounts, hipMemcpyDeviceToHost, st));
    if (N >= 2) RT_HIP(ctx, hipMemcpyAsync(&hostPlan, plan, sizeof hostPlan, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipStreamSynchronize(st));
    if (N >= 2) {
        ctx->rootBox = hostPlan.root;
        ctx->grid = hostPlan.grid;
        memcpy(&ctx->smallEdgeProduct, &hostPlan.smallEdgeProduct, 4);
        memcpy(&ctx->smallEdgeSum, &hostPlan.smallEdgeSum, 4);
    }
    RT_HIP(ctx, hipGetLastError());
    for (uint32_t k = 0; k < sc.nFinal; ++k)
        if (hostCounts[sc.finalCount[k]] != 0u)
            return rt_fail(ctx, RT_E_STATE, "BVH build: %u nodes were still unfinished after the %u / %u levels a bottom-up pass is given", hostCounts[sc.finalCount[k]], FIT_LEVELS, PASS_LEVELS);
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    ctx->buildMs = ms;
    ctx->built = true;
    return RT_OK;
}

template <typename T>
static int fetch(RtContext *ctx, T *dst, const T *src, size_t count)
{
    if (!dst || count == 0) return RT_OK;
    RT_HIP(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return RT_OK;
}

int rt_readback_bvh_impl(RtContext *ctx, uint32_t *morton, uint32_t *sortedIdx, RtAABB *triAABB, RtTriangle *tris,
                         RtInternalNode *inodes, RtLeafNode *leaves, RtAABB *bvhData)
{
    const uint32_t N = ctx->N;
    if (N == 0) return RT_OK;
    int rc;
    if ((rc = fetch(ctx, morton, ctx->dMorton, N))) return rc;
    if ((rc = fetch(ctx, sortedIdx, ctx->dSortedIdx, N))) return rc;
    if ((rc = fetch(ctx, triAABB, ctx->dTriAABB, N))) return rc;
    if ((rc = fetch(ctx, bvhData, ctx->dBvhData, N - 1))) return rc;
    RtInternalNode *dI = nullptr;
    RtLeafNode *dL = nullptr;
    RtTriangle *dT = nullptr;
    if (inodes || leaves) {
        RT_HIP(ctx, dalloc(&dI, (size_t)N));
        RT_HIP(ctx, dalloc(&dL, (size_t)N));
        if (N >= 2) {
            hipLaunchKernelGGL(k_wire_nodes, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, N, ctx->dLeft, ctx->dRight,
                               ctx->dParent, ctx->dLeafParent, dI, dL);
        } else {
            RT_HIP(ctx, hipMemsetAsync(dL, 0xFF, sizeof(RtLeafNode), ctx->stream));  // never linked: stays NullLeaf
        }
        if ((rc = fetch(ctx, inodes, dI, N - 1))) return rc;
        if ((rc = fetch(ctx, leaves, dL, N))) return rc;
    }
    if (tris) {
        RT_HIP(ctx, dalloc(&dT, (size_t)N));
        hipLaunchKernelGGL(k_wire_triangles, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, N, ctx->dIdx,
                           (const float *)ctx->dVerts, ctx->dMatIdx, ctx->dShadow, dT);
        if ((rc = fetch(ctx, tris, dT, N))) return rc;
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipGetLastError());
    if (dI) (void)hipFree(dI);
    if (dL) (void)hipFree(dL);
    if (dT) (void)hipFree(dT);
    return RT_OK;
}
