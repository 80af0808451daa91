// Closest-hit queries: the packed pair test, the cooperative leaf phase, the node walk of deep trees, and the sifting
// traversal of shallow trees (BVH::hit, aggregate.rs:324-401; List::hit, aggregate.rs:144-165).
// Each function names the reference code it computes (paths relative to the reference root).
#pragma once
#include "rrt_primitives.hpp"
#include "rrt_profile.hpp"

namespace rrt {




// ---- closest hit: BVH::hit (aggregate.rs:324-401) with List::hit (aggregate.rs:144-165) in the leaves ------------
struct Hit {
    int slot;  // -1 = miss
    float t;
    float u, v, det;  // triangle hits: the values Triangle::hit computed (reused by the hit properties)
};

// One primitive of a leaf against the ray: updates `h` when it is hit closer than h.t (strict, first visited wins ties).
template <uint32_t FEAT>
RRT_DEV void test_slot(const DScene& sc, uint32_t s, const Ray& r, float t_min, const Rng& key, uint32_t depth, Hit& h) {
    const cv4p cslots = sc.c_slots();
    const v4f q0 = cslots[3 * s], q1 = cslots[3 * s + 1], q2 = cslots[3 * s + 2];
    const uint32_t info = __float_as_uint(q1.w);
    const uint32_t kind = RRT_INFO_KIND(info);
    if (!(FEAT & (F_SPHERE | F_MEDIUM)) || kind == RRT_KIND_TRIANGLE) {
        float t, u, v, det;
        if (triangle_hit_uv(xyz(q0), xyz(q1), xyz(q2), (info & RRT_INFO_CULL) != 0, r, t_min, h.t, &t, &u, &v, &det)) {
            h.t = t;
            h.slot = (int)s;
            h.u = u;
            h.v = v;
            h.det = det;
        }
    } else if ((FEAT & F_SPHERE) && kind == RRT_KIND_SPHERE) {
        float t;
        if (sphere_hit(sc.c_spheres() + __float_as_uint(q2.w), r, t_min, h.t, &t)) {
            h.t = t;
            h.slot = (int)s;
        }
    } else if (FEAT & F_MEDIUM) {
        float t;
        const cmedium_p m = sc.c_media() + __float_as_uint(q2.w);
        const uint32_t tag = __float_as_uint(q0.w);  // the shape's index in scene order
        const bool found = (FEAT & F_NESTED) && m->boundary_kind == RRT_KIND_MEDIUM ? medium_hit_nested(sc, m, r, t_min, h.t, key, depth, tag, &t)
                                                                                     : medium_hit(sc, m, r, t_min, h.t, rng_keyed(key, depth, tag), &t);
        if (found) {
            h.t = t;
            h.slot = (int)s;
        }
    }
}

// Two consecutive triangles of a leaf at once.  The pair record interleaves the two triangles component by component
// (v0x_A, v0x_B, v0y_A, v0y_B, ...), so every multiply / add / subtract of Triangle::hit (shape.rs:335-372) is one packed
// two-lane f32 instruction; each lane of a packed operation rounds exactly like the scalar one, and the two results are
// accepted in leaf order (A, then B against the distance A may have shortened), so the outcome equals two single tests.
typedef float v2f __attribute__((ext_vector_type(2)));
RRT_DEV v2f splat2(float x) { return v2f{x, x}; }
struct PairEval {
    v2f t, u, v, det;
    bool ok_a, ok_b;  // every acceptance test of Triangle::hit except `t < t_max`
};
// Returns false when both determinants were rejected (nothing else is computed then).
RRT_DEV bool triangle_pair_eval(v4f q0, v4f q1, v4f q2, v4f q3, v4f q4, const Ray& r, float t_min, PairEval& pe) {
    const v2f v0x = q0.xy, v0y = q0.zw, v0z = q1.xy, e1x = q1.zw, e1y = q2.xy, e1z = q2.zw, e2x = q3.xy, e2y = q3.zw, e2z = q4.xy;
    const uint32_t info_a = __float_as_uint(q4.z), info_b = __float_as_uint(q4.w);
    const v2f dx = splat2(r.d.x), dy = splat2(r.d.y), dz = splat2(r.d.z);
    // p_vec = dir x edge_2 ; determinant = edge_1 . p_vec
    const v2f px = (dy * e2z) - (dz * e2y);
    const v2f py = (dz * e2x) - (dx * e2z);
    const v2f pz = (dx * e2y) - (dy * e2x);
    const v2f det = ((e1x * px) + (e1y * py)) + (e1z * pz);
    // the cull flags are wave-uniform in the uniform leaf walk; the common all-culled pair takes a scalar branch to the plain comparisons
    bool ok_a, ok_b;
    if (info_a & info_b & RRT_INFO_CULL) {
        asm volatile("");  // keeps the two arms apart (merged, they cost a per-lane select each)
        ok_a = !(det.x < RRT_EPS);
        ok_b = !(det.y < RRT_EPS);
    } else {
        ok_a = (info_a & RRT_INFO_CULL) ? !(det.x < RRT_EPS) : !(__builtin_fabsf(det.x) < RRT_EPS);
        ok_b = (info_b & RRT_INFO_CULL) ? !(det.y < RRT_EPS) : !(__builtin_fabsf(det.y) < RRT_EPS);
    }
    pe.ok_a = pe.ok_b = false;
    if (!(ok_a || ok_b)) return false;
    // 1 / determinant for both triangles: reciprocal estimate + one fused Newton step (see exact_rcp); when the residual of
    // either is not small (zero, denormal, huge or non-finite determinant) the whole wave takes the division instead
    const v2f r0 = v2f{__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
    const v2f e = __builtin_elementwise_fma(-det, r0, splat2(1.0f));
    v2f inv = __builtin_elementwise_fma(e, r0, r0);
    if (__builtin_expect(__ballot(!(__builtin_fabsf(e.x) <= RRT_RCP_RESIDUAL) || !(__builtin_fabsf(e.y) <= RRT_RCP_RESIDUAL)) != 0ull, 0))
        inv = v2f{1.0f / det.x, 1.0f / det.y};
    const v2f tx = splat2(r.o.x) - v0x, ty = splat2(r.o.y) - v0y, tz = splat2(r.o.z) - v0z;
    const v2f u = (((tx * px) + (ty * py)) + (tz * pz)) * inv;
    // q_vec = t_vec x edge_1
    const v2f qx = (ty * e1z) - (tz * e1y);
    const v2f qy = (tz * e1x) - (tx * e1z);
    const v2f qz = (tx * e1y) - (ty * e1x);
    const v2f v = (((dx * qx) + (dy * qy)) + (dz * qz)) * inv;
    const v2f uv = u + v;
    const v2f t = (((e2x * qx) + (e2y * qy)) + (e2z * qz)) * inv;
    // acceptance in the reference's order of tests; the comparisons are written so that NaNs fall the same way
    pe.ok_a = ok_a & !(u.x < 0.0f) & !(u.x > 1.0f) & !(v.x < 0.0f) & !(uv.x > 1.0f) & (t.x > t_min);
    pe.ok_b = ok_b & !(u.y < 0.0f) & !(u.y > 1.0f) & !(v.y < 0.0f) & !(uv.y > 1.0f) & (t.y > t_min);
    pe.t = t;
    pe.u = u;
    pe.v = v;
    pe.det = det;
    return true;
}
RRT_DEV void triangle_pair_hit(v4f q0, v4f q1, v4f q2, v4f q3, v4f q4, v4f q5, const Ray& r, float t_min, Hit& h) {
    PairEval pe;
    if (!triangle_pair_eval(q0, q1, q2, q3, q4, r, t_min, pe)) return;
    // A first, then B against the distance A may have shortened (leaf order)
    const bool take_a = pe.ok_a && pe.t.x < h.t;
    const bool take_b = pe.ok_b && pe.t.y < (take_a ? pe.t.x : h.t);
    if (take_a) {
        h.t = pe.t.x;
        h.slot = (int)__float_as_uint(q5.x);
        h.u = pe.u.x;
        h.v = pe.v.x;
        h.det = pe.det.x;
    }
    if (take_b) {
        h.t = pe.t.y;
        h.slot = (int)__float_as_uint(q5.y);
        h.u = pe.u.y;
        h.v = pe.v.y;
        h.det = pe.det.y;
    }
}


template <uint32_t FEAT>
RRT_DEV void test_pair(const DScene& sc, cv4p rec, const Ray& r, float t_min, const Rng& key, uint32_t depth, Hit& h) {
    const v4f q4 = rec[4], q5 = rec[5];
    const uint32_t kind_a = RRT_INFO_KIND(__float_as_uint(q4.z)), kind_b = RRT_INFO_KIND(__float_as_uint(q4.w));
    if (kind_a == RRT_KIND_TRIANGLE && kind_b == RRT_KIND_TRIANGLE) {
        triangle_pair_hit(rec[0], rec[1], rec[2], rec[3], q4, q5, r, t_min, h);
    } else {
        test_slot<FEAT>(sc, __float_as_uint(q5.x), r, t_min, key, depth, h);
        if (kind_b != RRT_KIND_NONE) test_slot<FEAT>(sc, __float_as_uint(q5.y), r, t_min, key, depth, h);
    }
}

// `stack` arguments below point at the calling lane's first entry of an LDS array laid out [entry][thread]; `stride` =
// threads per block.  Leaves are stored as pair records (kPairQuads x float4 per two primitives); on the device a leaf
// node's `a` is its first pair and the low bits of `b` its number of pairs.  The two primitives of a pair are in leaf
// order (A before B) but pairs need not be: every comparison below orders candidates by (distance, slot), and slots are
// numbered in leaf order.

// monotone map float -> uint (all non-NaN values), and back
RRT_DEV uint32_t ord_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
RRT_DEV float ord_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

// ---- closest hit for deep trees: wave-cooperative leaves ------------------------------------------------------------
// On a large mesh the lanes of a wave sit in different leaves of very different sizes (1 to 50+ primitives in the
// reference's trees), and a per-lane leaf loop makes the whole wave wait for every visit.  Here a lane that reaches a leaf
// only notes it (first pair, pair count) and pauses; when enough lanes are paused (or nobody can walk on) the wave
// flattens all noted (lane, pair) items into one list and tests them 64 at a time, every lane taking ANY owner's pair:
// it fetches the owner's ray through the lane crossbar, runs the ordinary pair test against the owner's closest distance,
// and reduces with a 64-bit minimum on (distance bits, slot) in the wave's LDS area — the smallest distance wins,
// equal distances go to the earlier slot, which is what testing the leaf front to back does.  The walk order of every
// lane is untouched, so results are those of closest_hit.
//
// `coop` is this wave's LDS area: 64 x u64 keys followed by 64 x (slot, det, u, v).

// Tests up to 64 listed (owner lane, pair position) items, one per lane, for coop_leaves().
template <uint32_t FEAT, bool CLEAN>
RRT_DEV void coop_group(const DScene& sc, const Ray& r, float t_min, const Rng& key, uint32_t depth, uint32_t first_pair, uint32_t n_pairs,
                        uint32_t first_pair2, const Hit& h, const uint32_t* items, uint32_t n_items, unsigned long long* best, v4f* payload) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool have = lane < n_items;
    const uint32_t item = have ? items[lane] : lane;
    const int owner = (int)(item & 63u);
    const uint32_t k = item >> 6;  // position of the pair in its leaf (clean trees: in the owner's two held leaves, one after the other)
    uint32_t o_pair = (uint32_t)__shfl((int)first_pair, owner) + k;
    if (CLEAN) {
        const uint32_t o_n = (uint32_t)__shfl((int)n_pairs, owner), o_first2 = (uint32_t)__shfl((int)first_pair2, owner);
        if (k >= o_n) o_pair = o_first2 + (k - o_n);
    }
    Ray ro;
    ro.o = mk3(__shfl(r.o.x, owner), __shfl(r.o.y, owner), __shfl(r.o.z, owner));
    ro.d = mk3(__shfl(r.d.x, owner), __shfl(r.d.y, owner), __shfl(r.d.z, owner));
    const float o_t = __shfl(h.t, owner);
    Rng okey = key;
    uint32_t odepth = depth;
    if (FEAT & F_MEDIUM) {  // the medium's distance draw is keyed by the owner's stream and depth
        okey.s0 = (uint32_t)__shfl((int)key.s0, owner);
        okey.s1 = (uint32_t)__shfl((int)key.s1, owner);
        odepth = (uint32_t)__shfl((int)depth, owner);
    }
    Hit hh;
    hh.slot = -1;
    hh.t = o_t;
    if (CLEAN && __shfl(h.slot, owner) >= 0) hh.t = ord_float(ord_bits(o_t) + 1u);  // an equal distance may still win on shape order
    hh.u = hh.v = hh.det = 0.0f;
    unsigned long long mykey = ~0ull;
    if (have) {
        const cv4p rec = sc.c_pairs() + kPairQuads * o_pair;
        test_pair<FEAT>(sc, rec, ro, t_min, okey, odepth, hh);
        if (hh.slot >= 0) {
            uint32_t pos = (uint32_t)hh.slot;
            if (CLEAN) {
                const v4f q5 = rec[5];  // (slot A, slot B, shape A, shape B)
                pos = __float_as_uint(pos == __float_as_uint(q5.x) ? q5.z : q5.w);
            }
            mykey = ((unsigned long long)ord_bits(hh.t) << 32) | (unsigned long long)(pos + 1u);
            atomicMin(&best[owner], mykey);
        }
    }
    __threadfence_block();
    if (hh.slot >= 0 && best[owner] == mykey) payload[owner] = v4f{__uint_as_float((uint32_t)hh.slot), hh.det, hh.u, hh.v};
    __threadfence_block();
}

// The cooperative leaf phase: every pending lane stands at a leaf (first_pair, n_pairs); the wave lists all (lane, pair) items
// - position by position, compacted with a ballot and a prefix count, no cross-lane shuffles - and tests them 64 at a time, any
// lane testing any owner's pair.  Must be called by all 64 lanes.  Clean trees: a lane may bring a second leaf (first_pair2, n_pairs2;
// n_pairs2 = 0 without one), and `hprim` is the scene position + 1 of the shape its hit so far is on (0 without a hit): kept by the
// caller from phase to phase.
template <uint32_t FEAT, bool CLEAN>
RRT_DEV void coop_leaves(const DScene& sc, const Ray& r, float t_min, const Rng& key, uint32_t depth, bool pending, uint32_t first_pair, uint32_t n_pairs,
                         uint32_t first_pair2, uint32_t n_pairs2, Hit& h, uint32_t& hprim, unsigned char* coop) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* best = (unsigned long long*)coop;
    v4f* payload = (v4f*)(coop + 64 * 8);
    uint32_t* items = (uint32_t*)(coop + 64 * 8 + 64 * 16);  // 128 places: a group being filled plus one position's worth of items
    // Ties: in the reference's tree the leaf visited first wins, so what the owner already holds gets position 0; in a clean
    // tree the shape that is first in scene order wins wherever it sits, so positions are shape numbers throughout.
    const unsigned long long init = ((unsigned long long)ord_bits(h.t) << 32) | (CLEAN ? hprim : 0u);
    const uint32_t n_all = n_pairs + (CLEAN ? n_pairs2 : 0u);
    best[lane] = init;
    __threadfence_block();
    uint32_t count = 0;
    for (uint32_t k = 0;; ++k) {
        const bool has = pending && k < n_all;
        const unsigned long long bal = RRT_BALLOT(has);
        if (bal == 0ull) break;
        if (has) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            items[count + rank] = (k << 6) | lane;
        }
        count += (uint32_t)__popcll(bal);
        __threadfence_block();
        if (count >= 64u) {
            coop_group<FEAT, CLEAN>(sc, r, t_min, key, depth, first_pair, n_pairs, first_pair2, h, items, 64u, best, payload);
            count -= 64u;
            const uint32_t moved = lane < count ? items[64u + lane] : 0u;
            __threadfence_block();
            if (lane < count) items[lane] = moved;
            __threadfence_block();
        }
    }
    if (count != 0u) coop_group<FEAT, CLEAN>(sc, r, t_min, key, depth, first_pair, n_pairs, first_pair2, h, items, count, best, payload);
    if (pending && best[lane] != init) {
        const v4f w = payload[lane];
        h.t = ord_float((uint32_t)(best[lane] >> 32));
        h.slot = (int)__float_as_uint(w.x);
        h.det = w.y;
        h.u = w.z;
        h.v = w.w;
        if (CLEAN) hprim = (uint32_t)best[lane];
    }
}

// One lane's walk through a deep tree, kept between steps so that callers can interleave the lanes of a wave freely.
struct Walk {
    uint32_t sp, cur;
    bool running;  // has a node to visit
    bool pending;  // stands at a leaf (first_pair, n_pairs) that the next cooperative phase tests
    uint32_t first_pair, n_pairs;
    uint32_t held, held2;  // clean trees: the links of up to two leaves reached but not yet tested (0: none), see walk_step
    uint32_t hprim;        // clean trees: scene position + 1 of the shape the hit so far is on (ties go by scene order, see coop_leaves)
    f3 inv;
    uint32_t down;  // bit c: the ray runs towards smaller coordinates on axis c
    Hit h;
};
// A lane's stack of pending links.  The first lds_entries places sit in LDS ([entry][lane]); places beyond them - reached only
// on unusually long descents - sit in the lane's private memory, so that the LDS request (and with it the number of waves a
// compute unit holds) follows the common case and not the tree's worst case.
struct WalkStack {
    uint32_t* lds;
    uint32_t stride, lds_entries;
    uint32_t* beyond;
};
RRT_DEV void stack_push(const WalkStack& s, uint32_t& sp, uint32_t link) {
    if (sp < s.lds_entries) s.lds[sp * s.stride] = link;
    else s.beyond[sp - s.lds_entries] = link;
    ++sp;
}
RRT_DEV uint32_t stack_pop(const WalkStack& s, uint32_t& sp) {
    --sp;
    return sp < s.lds_entries ? s.lds[sp * s.stride] : s.beyond[sp - s.lds_entries];
}
template <int AABB_MODE>
RRT_DEV void walk_begin(const DScene& sc, Walk& w, bool active, const Ray& r, float t_max) {
    w.h.slot = -1;
    w.h.t = t_max;
    w.h.u = w.h.v = w.h.det = 0.0f;
    w.sp = 0;
    w.cur = sc.root_link;  // 0 for the reference's trees
    w.pending = false;
    w.first_pair = w.n_pairs = 0;
    w.held = w.held2 = w.hprim = 0u;
    w.running = active && sc.n_nodes != 0 && w.cur != kLinkNone;  // kLinkNone: a clean tree whose only leaf is empty
    w.inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    if (AABB_MODE == kAabbClean) {
        // A ray that runs along an axis (a direction component of exactly zero) has an infinite inverse there, and the clean walk's
        // fused plane test q * (step * inv) + (origin - o) * inv would turn into inf - inf: no answer, the axis would cull nothing
        // and the ray would visit every node along its plane.  2^96 in its place keeps the test's meaning - inside the slab the
        // interval is (-huge, +huge), beside it the interval lies beyond every distance a scene has - and stays finite.
        const float cap = 0x1p96f;
        if (__builtin_isinf(w.inv.x)) w.inv.x = __builtin_copysignf(cap, w.inv.x);
        if (__builtin_isinf(w.inv.y)) w.inv.y = __builtin_copysignf(cap, w.inv.y);
        if (__builtin_isinf(w.inv.z)) w.inv.z = __builtin_copysignf(cap, w.inv.z);
    }
    w.down = (w.inv.x < 0.0f ? 1u : 0u) | (w.inv.y < 0.0f ? 2u : 0u) | (w.inv.z < 0.0f ? 4u : 0u);
}
// Visits one node (BVH::hit, aggregate.rs:343-395): an inner node that passes its test pushes one child and descends into the
// other, a leaf that passes pauses the walk, anything else pops.
template <int AABB_MODE>
RRT_DEV void walk_step(const DScene& sc, Walk& w, const Ray& r, float t_min, const WalkStack& stack) {
    const cv4p cnodes = sc.c_nodes();
    if (AABB_MODE == kAabbClean) {
        // One 64-byte fetch brings the boxes of up to four children: only children the ray meets are ever fetched themselves,
        // and leaves are not fetched at all (their pair range sits in the link).  Children are visited nearest entry first;
        // the result does not depend on the order (ties go by scene order, see coop_leaves).
        // A lane that reaches a leaf does not stop for it: it keeps the leaf's link ("held"), and a second leaf's too, and walks on,
        // and stops only at the THIRD leaf (or when its walk is over).  The wave's cooperative phases test the held leaves of ALL lanes, walking or not,
        // so a phase serves more lanes and fewer lanes stand idle waiting for one.  The nodes visited meanwhile are culled with
        // the closest distance known before the held leaf's test - a few more visits, the same result: in a clean tree the
        // answer does not depend on the order of the tests (nearest hit, ties by scene order).
        if (w.cur & kLinkLeaf) {
            if (w.held2 == 0u) {
                if (w.held == 0u) w.held = w.cur;
                else w.held2 = w.cur;
                if (w.sp == 0) w.running = false;
                else w.cur = stack_pop(stack, w.sp);
                return;
            }
            w.pending = true;  // (w.cur keeps the third leaf's link)
            w.running = false;
            return;
        }
        // One 64-byte node in four loads: the grid (origin, steps), the children's planes as bytes on it, the links.
        const uint32_t base = 4u * w.cur;
        const v4f g0 = cnodes[base], g1 = cnodes[base + 1u], g2 = cnodes[base + 2u], g3 = cnodes[base + 3u];
        const uint32_t expo = __float_as_uint(g0.w);
        // t(plane) = (origin + q * step - o) * inv = q * (step * inv) + (origin - o) * inv: one fused multiply-add per plane.  Its
        // rounding (a few ulp of the ray's distance to the node) is a hundredth of the padding the stored boxes carry.
        const f3 a = mk3(__uint_as_float((expo & 255u) << 23) * w.inv.x, __uint_as_float(((expo >> 8) & 255u) << 23) * w.inv.y,
                         __uint_as_float(((expo >> 16) & 255u) << 23) * w.inv.z);
        const f3 b = mk3((g0.x - r.o.x) * w.inv.x, (g0.y - r.o.y) * w.inv.y, (g0.z - r.o.z) * w.inv.z);
        // The ray's direction signs pick, per axis, which of the two plane dwords is the near one: no per-axis minimum and maximum.
        const bool dx = (w.down & 1u) != 0u, dy = (w.down & 2u) != 0u, dz = (w.down & 4u) != 0u;
        const uint32_t lox = __float_as_uint(g1.x), hix = __float_as_uint(g1.y), loy = __float_as_uint(g1.z), hiy = __float_as_uint(g1.w),
                       loz = __float_as_uint(g2.x), hiz = __float_as_uint(g2.y);
        const uint32_t nx = dx ? hix : lox, fx = dx ? lox : hix, ny = dy ? hiy : loy, fy = dy ? loy : hiy, nz = dz ? hiz : loz, fz = dz ? loz : hiz;
        const uint32_t links[4] = {__float_as_uint(g2.z), __float_as_uint(g2.w), __float_as_uint(g3.x), __float_as_uint(g3.y)};
        // (order, link) of the children the ray meets, the others (kLinkNone) last; the exchanges compare the 32-bit order alone - a
        // 64-bit compare on (order, link) costs a quarter-rate instruction and a wait per exchange, and which of two children at the
        // same distance comes first changes nothing in a clean tree
        uint32_t ord[4], lnk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t link = links[k];
            // NaN (0 * inf, inf - inf: the ray lies in a plane of the box or runs along an axis) is dropped by the maxima and minima:
            // that plane then constrains nothing, which can only add visits
            const float entry = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((nx >> (8 * k)) & 255u), a.x, b.x),
                                                                __builtin_fmaf((float)((ny >> (8 * k)) & 255u), a.y, b.y)),
                                                __builtin_fmaxf(__builtin_fmaf((float)((nz >> (8 * k)) & 255u), a.z, b.z), t_min));
            const float leave = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((fx >> (8 * k)) & 255u), a.x, b.x),
                                                                __builtin_fmaf((float)((fy >> (8 * k)) & 255u), a.y, b.y)),
                                                __builtin_fminf(__builtin_fmaf((float)((fz >> (8 * k)) & 255u), a.z, b.z), w.h.t));
            const bool met = entry <= leave && link != kLinkNone;
            // entry >= t_min: with a positive lower bound (a constant in the render kernels) the float's own bits order correctly
            const uint32_t order = t_min > 0.0f ? __float_as_uint(entry) : ord_bits(entry);
            ord[k] = met ? order : 0xffffffffu;
            lnk[k] = met ? link : kLinkNone;
        }
#define RRT_ORDER(i, j)                                    \
    {                                                      \
        const bool swap_ = ord[j] < ord[i];                \
        const uint32_t oi_ = ord[i], li_ = lnk[i];         \
        ord[i] = swap_ ? ord[j] : oi_;                     \
        lnk[i] = swap_ ? lnk[j] : li_;                     \
        ord[j] = swap_ ? oi_ : ord[j];                     \
        lnk[j] = swap_ ? li_ : lnk[j];                     \
    }
        RRT_ORDER(0, 1) RRT_ORDER(2, 3) RRT_ORDER(0, 2) RRT_ORDER(1, 3) RRT_ORDER(1, 2)
#undef RRT_ORDER
        if (lnk[0] == kLinkNone) {
            if (w.sp == 0) w.running = false;
            else w.cur = stack_pop(stack, w.sp);
            return;
        }
        w.cur = lnk[0];
#pragma unroll
        for (int k = 3; k >= 1; --k)
            if (lnk[k] != kLinkNone) stack_push(stack, w.sp, lnk[k]);
        return;
    }
    const v4f n0 = cnodes[4 * w.cur];
    const v4f n1 = cnodes[4 * w.cur + 1];
    bool inside;
    if (sc.is_list) inside = true;
    else if (AABB_MODE == RRT_AABB_TIGHT) {
        const v4f b0 = cnodes[4 * w.cur + 2], b1 = cnodes[4 * w.cur + 3];
        inside = box_overlap(xyz(b0), xyz(b1), r.o, w.inv, t_min, w.h.t) && aabb_reference(xyz(n0), xyz(n1), r.o, w.inv, t_min, w.h.t);
    } else inside = aabb_reference(xyz(n0), xyz(n1), r.o, w.inv, t_min, w.h.t);
    const uint32_t a = __float_as_uint(n0.w), b = __float_as_uint(n1.w);
    if (inside && !(b & RRT_LEAF_FLAG)) {
        // aggregate.rs:378-392: with dir[axis] < 0 the left child (cur+1) is popped first, otherwise the right one
        const float dax = b == 0u ? r.d.x : (b == 1u ? r.d.y : r.d.z);
        const uint32_t left = w.cur + 1u, right = a;
        const bool left_first = dax < 0.0f;
        stack_push(stack, w.sp, left_first ? right : left);
        w.cur = left_first ? left : right;
    } else if (inside) {
        w.pending = true;  // the leaf is tested in the next cooperative phase; the walk resumes after it
        w.running = false;
        w.first_pair = a;
        w.n_pairs = b & ~RRT_LEAF_FLAG;
    } else if (w.sp == 0) {
        w.running = false;
    } else {
        w.cur = stack_pop(stack, w.sp);
    }
}
// Clean trees: the pair range of a leaf link.
RRT_DEV uint32_t link_first_pair(uint32_t link) { return link & 0x0fffffffu; }
RRT_DEV uint32_t link_pairs(uint32_t link) { return ((link >> 28) & 7u) + 1u; }
// Clean trees, after a cooperative phase that tested this lane's held leaves: a lane that stood at a third leaf now holds
// that one and walks on.  Returns false when nothing is left to do for the query (no node to visit, no leaf held).
RRT_DEV bool walk_after_held(Walk& w, const WalkStack& stack) {
    w.held = w.held2 = 0u;
    if (w.pending) {
        w.held = w.cur;
        w.pending = false;
        if (w.sp != 0) {
            w.cur = stack_pop(stack, w.sp);
            w.running = true;
        }
    }
    return w.running || w.held != 0u;
}
// After the cooperative phase that tested this lane's leaf.
RRT_DEV void walk_resume(Walk& w, const WalkStack& stack) {
    w.pending = false;
    if (w.sp != 0) {
        w.cur = stack_pop(stack, w.sp);
        w.running = true;
    }
}

// Recomputes (u, v, det) the way Triangle::get_hit_properties does (shape.rs:384-400) for callers without a Hit.
RRT_DEV void triangle_uvdet(f3 v0, f3 e1, f3 e2, const Ray& r, float* u, float* v, float* det) {
    f3 p_vec = cross3(r.d, e2);
    float d = dot3(e1, p_vec);
    float inv_det = 1.0f / d;
    f3 t_vec = r.o - v0;
    *u = dot3(t_vec, p_vec) * inv_det;
    f3 q_vec = cross3(t_vec, e1);
    *v = dot3(r.d, q_vec) * inv_det;
    *det = d;
}
// ---- closest hit for shallow trees: sift, then test densely ------------------------------------------------------------
// The reference's trees for small scenes are one or a few big leaves (its SAH rarely splits them), so every ray would run
// the full test against every primitive.  Here a lane that reaches a leaf pauses; once no lane is walking, the wave
//  1. sifts: each paused lane tests its ray against the padded box of every pair of its leaf (wave-uniform pair index, box
//     in scalar registers) and appends an item (pair, lane) to a list in LDS for every box its ray overlaps.  A pair whose
//     box the ray misses within (t_min, closest so far] holds nothing the ray can hit, so nothing is lost;
//  2. tests densely: the items are handed out 64 at a time, any lane taking any owner's item: it fetches the owner's ray
//     through the lane crossbar, runs the exact pair test and posts (distance, slot) with a 64-bit LDS
//     minimum - the smallest distance wins and equal distances go to the earlier primitive, which is what testing the
//     leaf front to back with a strict comparison does (aggregate.rs:144-165).
// The walk order of every lane is the reference's, so results equal closest_hit.
// `area` is this wave's LDS: 64 x u64 keys, then kSiftCap x u32 items (then 64 x kHarvestLights floats when harvesting).
//
// HARVEST (single-leaf scenes with a few lights).  After a diffuse bounce the reference evaluates the light mixture's
// density for the scattered ray, which costs one more intersection test per light (Shape::pdf, shape.rs:141-147, 441-458)
// - against the very primitives the NEXT closest-hit query of that same ray tests anyway.  With HARVEST the query also
// reports, per light, the distance at which the ray meets the light's own primitive (any distance in (t_min, t_max), whether
// or not something nearer hides it, and whether or not the leaf's box test let the ray in: Shape::pdf asks the shape, not
// the tree), and the caller finishes the previous bounce's throughput from that.  Same tests, same numbers, run once.

template <uint32_t FEAT, bool HARVEST, uint32_t CAP = kSiftCap>
RRT_DEV void sift_flush(const DScene& sc, const Ray& r, float t_min, float t_max, const Rng& key, uint32_t depth, const uint32_t* items, uint32_t n_items,
                        unsigned long long* best) {
    constexpr uint32_t kLaneShift = HARVEST ? 7 : 6;  // HARVEST items carry one more bit: the owner only wants its lights
    float* light_t = (float*)((unsigned char*)best + sift_bytes(CAP));
    const uint32_t lane = threadIdx.x & 63u;
    const cv4p cpairs = sc.c_pairs();
    __threadfence_block();
    for (uint32_t base = 0; base < n_items; base += 64u) {
        const uint32_t idx = base + lane;
        const bool have = idx < n_items;
        const uint32_t item = have ? items[idx] : lane;
        const int owner = (int)(item & 63u);
        const uint32_t pair = item >> kLaneShift;
        const bool lights_only = HARVEST && (item & 64u) != 0u;
        Ray ro;
        ro.o = mk3(__shfl(r.o.x, owner), __shfl(r.o.y, owner), __shfl(r.o.z, owner));
        ro.d = mk3(__shfl(r.d.x, owner), __shfl(r.d.y, owner), __shfl(r.d.z, owner));
        Rng okey = key;
        uint32_t odepth = depth;
        if (FEAT & F_MEDIUM) {  // the medium's distance draw is keyed by the owner's stream and depth
            okey.s0 = (uint32_t)__shfl((int)key.s0, owner);
            okey.s1 = (uint32_t)__shfl((int)key.s1, owner);
            odepth = (uint32_t)__shfl((int)depth, owner);
        }
        if (have) {
            const cv4p rec = cpairs + kPairQuads * pair;
            const v4f q4 = rec[4];
            const uint32_t kind_a = RRT_INFO_KIND(__float_as_uint(q4.z)), kind_b = RRT_INFO_KIND(__float_as_uint(q4.w));
            const v4f q5 = rec[5];
            const uint32_t light_a = HARVEST ? RRT_PAIR_LIGHT(__float_as_uint(q4.z)) : 0u, light_b = HARVEST ? RRT_PAIR_LIGHT(__float_as_uint(q4.w)) : 0u;
            bool found = false, second = false, det_neg = false;
            float t_found = t_max;
            if (kind_a == RRT_KIND_TRIANGLE && kind_b == RRT_KIND_TRIANGLE) {
                PairEval pe;
                if (triangle_pair_eval(rec[0], rec[1], rec[2], rec[3], q4, ro, t_min, pe)) {
                    const bool ok_a = pe.ok_a && pe.t.x < t_max, ok_b = pe.ok_b && pe.t.y < t_max;
                    if (HARVEST && (light_a | light_b) != 0u) {  // Triangle::pdf's own hit test (shape.rs:446): this one
                        if (light_a != 0u && ok_a) light_t[(uint32_t)owner * kHarvestLights + light_a - 1u] = pe.t.x;
                        if (light_b != 0u && ok_b) light_t[(uint32_t)owner * kHarvestLights + light_b - 1u] = pe.t.y;
                    }
                    second = ok_b && !(ok_a && !(pe.t.y < pe.t.x));  // A keeps an equal distance (leaf order)
                    found = ok_a || ok_b;
                    t_found = second ? pe.t.y : pe.t.x;
                    det_neg = (second ? pe.det.y : pe.det.x) < 0.0f;
                }
            } else {
                Hit hh;
                hh.slot = -1;
                hh.t = t_max;
                hh.u = hh.v = hh.det = 0.0f;
                test_slot<FEAT>(sc, __float_as_uint(q5.x), ro, t_min, okey, odepth, hh);
                if (HARVEST && light_a != 0u && hh.slot >= 0) light_t[(uint32_t)owner * kHarvestLights + light_a - 1u] = hh.t;  // Sphere::pdf, shape.rs:142
                if (kind_b != RRT_KIND_NONE) {
                    const int before = hh.slot;
                    const float t_before = hh.t;
                    test_slot<FEAT>(sc, __float_as_uint(q5.y), ro, t_min, okey, odepth, hh);
                    second = hh.slot >= 0 && (before < 0 || hh.t < t_before);
                }
                found = hh.slot >= 0;
                t_found = hh.t;
                det_neg = hh.det < 0.0f;
            }
            if (found && !lights_only) {
                const uint32_t slot = __float_as_uint(second ? q5.y : q5.x);
                const uint32_t low = ((slot + 1u) << 1) | (det_neg ? 1u : 0u);
                atomicMin(&best[owner], ((unsigned long long)ord_bits(t_found) << 32) | (unsigned long long)low);
            }
        }
    }
    __threadfence_block();
}

// ---- the grazing guard of the box shortcuts -----------------------------------------------------------------------------
// The padded boxes of the sifting query (pair boxes, and in TIGHT mode the primitive boxes of a shallow tree's nodes) drop a
// primitive only when the ray's span misses its box.  That equals the reference - which tests every shape of a visited leaf,
// aggregate.rs:144-165 - as long as every hit Triangle::hit (shape.rs:335-372) can accept lies inside the padded box.  Its
// rounding lets in plane crossings OUTSIDE the outline, by a distance that grows as 1 / sin(angle between ray and plane).  With
// u = 2^-24, a the angle between ray and plane, phi the angle between the triangle's two edges at v0, r the ratio of their lengths
// (>= 1), L the longer one, every operation of the unfused f32 test rounded once:
//   crossing point, in the plane   <= u / (sin a * sin phi) * (1 + 2 r) * (14 |o - v0| + 6 L)
//   hit distance, along the ray    <= u / (sin a * sin phi) * (7 |o - v0| + 6 (|o - v0| + L))     =: ds
// so the reported point o + t d lies within u * C * reach / (sin a * sin phi) of the triangle in its plane, C = (1 + 2 r) (14 + 6 l)
// + 7 + 6 (1 + l), and within ds * sin a <= (13 + 6 l) u * reach / sin phi of the plane, where `reach` = |o - centre| + rho bounds
// |o - v0| (a sphere (centre, rho) holds every triangle vertex of the scene) and l = min(L / rho, 2) bounds L / reach.  A triangle's box
// is padded by `pad_p` along its plane and by `pad_n` across it, so both shortcuts are exact for a ray with
//   |d . n| * sin phi / C  >=  |d| * reach * u / (0.95 pad_p)      and      reach <= 0.95 pad_n * sin phi / ((13 + 6 l) u)
// (a twentieth of the padding is left to the box test's own rounding, a few u of the coordinates).  The host stores, per distinct
// plane direction of the scene's triangles, G = n * min(sin phi / C) * 0.95 pad_p / (1.05 u) - the 1.05 covers directions merged into
// one entry although they differ in the sixth digit - and the smallest right-hand side of the second condition, over 1.05.  A ray with |d . G| < |d| * reach
// for ANY entry, or too long a reach, takes no shortcut at all: every box test of its query is made to pass, so it tests what the
// reference tests.  With the shallow trees' paddings that is about one ray in two thousand per plane direction.  Returns true for
// such a ray (and for rays with non-finite or vanishing directions).
// The table: (centre.xyz, rho); (gx, gy, gz, n) - the lengths of the entries along the three axes (infinite: none; most scenes'
// walls) and the number of general entries, an even number (made up with an entry no ray is close to: (inf, 0, 0)) that may be
// zero - the table still has room for two, which are fetched with the header -; (longest reach, square of the reach of a ray that starts at the
// camera or on a primitive, flag, the same for a ray that starts on a primitive: every ray of a path but its first); then the general entries, two to two float4: (x, x, y, y), (z, z, -, -).
// ANY_ORIGIN = false (the render kernels, whose rays start at the camera or on a primitive) takes one of those two reaches
// instead of working out the ray's own, unless the flag is set: the host sets it when the one reach is too long to be useful (a
// small mesh in a wide scene) or longer than the longest reach.
// (A direction too short for its squared length to be a normal float cannot hit a triangle at all: the reference's determinant
// test turns it down, and the host checks that the scene's triangles are small enough for that argument.)
RRT_DEV float min3_abs(float m, float a, float b) {  // min(m, |a|, |b|), NaNs dropped
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
// d . Ga and d . Gb for two entries stored side by side (Ga.x, Gb.x, Ga.y, Gb.y | Ga.z, Gb.z, -, -): three packed instructions
RRT_DEV v2f dot_two(f3 d, v4f xy, v4f z) {
    return __builtin_elementwise_fma(splat2(d.z), z.xy, __builtin_elementwise_fma(splat2(d.y), xy.zw, splat2(d.x) * xy.xy));
}
template <int AABB_MODE, bool ANY_ORIGIN>
RRT_DEV bool grazes_a_plane(const DScene& sc, const Ray& r, bool from_camera) {
    if (AABB_MODE == kAabbExhaustive) return false;  // (that mode has no shortcut to guard)
    // (the header and the first two general entries are fetched together: the table always has room for two)
    const cv4p g = sc.c_guards();
    const v4f h0 = g[0], h1 = g[1], h2 = g[2], Gxy = g[3], Gz = g[4];
    const uint32_t n = __float_as_uint(h1.w);
    float m = min3_abs(__builtin_fabsf(r.d.x) * h1.x, r.d.y * h1.y, r.d.z * h1.z);
    if (n != 0u) {  // (none in a scene whose triangles all lie across an axis: floors and walls)
        const v2f first = dot_two(r.d, Gxy, Gz);
        m = min3_abs(m, first.x, first.y);
        for (uint32_t j = 2; j < n; j += 2u) {
            const v2f more = dot_two(r.d, g[3u + j], g[4u + j]);
            m = min3_abs(m, more.x, more.y);
        }
    }
    const float dd = __builtin_fmaf(r.d.z, r.d.z, __builtin_fmaf(r.d.y, r.d.y, r.d.x * r.d.x));
    if (!ANY_ORIGIN && __float_as_uint(h2.z) == 0u) return !(m * m > dd * (from_camera ? h2.y : h2.w));
    const float ox = r.o.x - h0.x, oy = r.o.y - h0.y, oz = r.o.z - h0.z;
    const float reach = __builtin_amdgcn_sqrtf(__builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox))) + h0.w;
    const float lim2 = (dd * (reach * reach)) * 1.0001f;  // (the roundings of this function's own arithmetic)
    return !(m * m > lim2) || !(reach <= h2.x);
}

template <int AABB_MODE, uint32_t FEAT, bool HARVEST = false, uint32_t CAP = kSiftCap, bool WANT_UV = true, bool ANY_ORIGIN = true>
RRT_DEV Hit closest_hit_sift(const DScene& sc, bool active, const Ray& r, float t_min, float t_max, const Rng& key, uint32_t depth, uint32_t* stack,
                             uint32_t stride, unsigned char* area, float* harvest = nullptr) {
    // Must be called by all 64 lanes of the wave (the dense phase addresses lanes by number); lanes without a ray pass
    // active = false and only lend a hand.  HARVEST: the tree is a single leaf; harvest[i] receives the distance at which the
    // ray meets light i's primitive, or -1.
    Hit h;
    h.slot = -1;
    h.t = t_max;
    h.u = h.v = h.det = 0.0f;
    if (sc.n_nodes == 0) return h;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* best = (unsigned long long*)area;
    uint32_t* items = (uint32_t*)(area + 64 * 8);
    // a ray that runs too close to a triangle's plane for the padded boxes to be trusted (see grazes_a_plane) gets NaN on all
    // three axes below: every pair box is then met, and so is every node's primitive box
    const bool graze = grazes_a_plane<AABB_MODE, ANY_ORIGIN>(sc, r, depth == 0u);
    const f3 inv = exact_rcp3(r.d);  // = 1.0f / r.d, bit for bit (see exact_rcp)
    // for the pair boxes: centre * finv + noi (see pair_box_overlap_ch); an axis whose product origin * inv is not finite (zero or
    // denormal direction component) gets NaN, which the interval test ignores: that axis then constrains nothing
    f3 finv = inv, noi = mk3(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
    if (graze || !(__builtin_fabsf(noi.x) < __builtin_inff())) { finv.x = 0.0f; noi.x = __builtin_nanf(""); }
    if (graze || !(__builtin_fabsf(noi.y) < __builtin_inff())) { finv.y = 0.0f; noi.y = __builtin_nanf(""); }
    if (graze || !(__builtin_fabsf(noi.z) < __builtin_inff())) { finv.z = 0.0f; noi.z = __builtin_nanf(""); }
    const v2f_ sx = {-__builtin_fabsf(finv.x), __builtin_fabsf(finv.x)}, sy = {-__builtin_fabsf(finv.y), __builtin_fabsf(finv.y)},
               sz = {-__builtin_fabsf(finv.z), __builtin_fabsf(finv.z)};  // see pair_box_overlap_ch
    SiftProbe probe;
    const cv4p cnodes = sc.c_nodes();
    const cv4p cpairs = sc.c_pairs();
    uint32_t sp = 0, cur = 0;
    bool running = active, pending = false;
    bool lights_only = false;  // HARVEST: the leaf's box test failed, so only the lights' own primitives are of interest
    uint32_t first_pair = 0, n_pairs = 0;
    float* light_t = (float*)(area + sift_bytes(CAP));
    if (HARVEST)
        for (uint32_t i = 0; i < kHarvestLights; ++i) harvest[i] = -1.0f;
    for (;;) {
        if (running) {
            const uint32_t node = HARVEST ? 0u : cur;  // (a harvesting query's tree is its root: the node comes by scalar loads)
            const v4f n0 = cnodes[4 * node];
            const v4f n1 = cnodes[4 * node + 1];
            bool inside;
            if (sc.is_list) inside = true;
            else if (AABB_MODE == RRT_AABB_TIGHT && !HARVEST) {
                // (a harvesting query's tree is one leaf: a ray that misses the leaf's primitive box misses every pair box in it,
                // so the test would decide nothing that the sifting loop does not decide anyway)
                const v4f b0 = cnodes[4 * node + 2], b1 = cnodes[4 * node + 3];
                inside = (graze || box_overlap(xyz(b0), xyz(b1), r.o, inv, t_min, h.t)) && aabb_reference(xyz(n0), xyz(n1), r.o, inv, t_min, h.t);
            } else inside = aabb_reference(xyz(n0), xyz(n1), r.o, inv, t_min, h.t);
            const uint32_t a = __float_as_uint(n0.w), b = __float_as_uint(n1.w);
            if (!HARVEST && inside && !(b & RRT_LEAF_FLAG)) {
                // aggregate.rs:378-392: with dir[axis] < 0 the left child (cur+1) is popped first, otherwise the right one
                const float dax = b == 0u ? r.d.x : (b == 1u ? r.d.y : r.d.z);
                const uint32_t left = cur + 1u, right = a;
                const bool left_first = dax < 0.0f;
                stack[sp * stride] = left_first ? right : left;
                ++sp;
                cur = left_first ? left : right;
            } else if (inside || HARVEST) {
                pending = true;  // the leaf is tested in the next sift phase; the walk resumes after it
                running = false;
                lights_only = !inside;
                first_pair = a;
                n_pairs = b & ~RRT_LEAF_FLAG;
            } else if (sp == 0) {
                running = false;
            } else {
                --sp;
                cur = stack[sp * stride];
            }
        }
        if (!HARVEST && RRT_BALLOT(running) != 0ull) continue;  // (a harvesting query's walk is that one node)
        const unsigned long long waiting = RRT_BALLOT(pending);
        if (waiting == 0ull) break;
        best[lane] = (unsigned long long)ord_bits(h.t) << 32;  // position 0: what the owner already holds wins ties
        if (HARVEST)
            for (uint32_t i = 0; i < kHarvestLights; ++i) light_t[lane * kHarvestLights + i] = -1.0f;
        uint32_t n_items = 0;
        unsigned long long todo = waiting;
        unsigned long long direct_best = ~0ull;  // this phase's nearest directly tested primitive, as the dense groups' key
        probe.sift_begins((uint32_t)__popcll(waiting));
        while (todo != 0ull) {
            // the paused lanes are grouped by leaf (one group for the single-leaf trees of the small scenes)
            uint32_t leaf_first, leaf_word;
            if (HARVEST) {  // one leaf for everybody: its record comes by scalar loads
                leaf_first = __float_as_uint(cnodes[0].w);
                leaf_word = __float_as_uint(cnodes[1].w) & ~RRT_LEAF_FLAG;
            } else {
                const int leader = __ffsll((long long)todo) - 1;
                leaf_first = (uint32_t)__builtin_amdgcn_readlane((int)first_pair, leader);
                leaf_word = (uint32_t)__builtin_amdgcn_readlane((int)n_pairs, leader);
            }
            const uint32_t leaf_pairs = leaf_word & ~kLeafDirect;
            const bool mine = HARVEST ? pending : pending && first_pair == leaf_first;
            // Lanes that take no part get an upper bound of -infinity, which no box interval reaches: the loop then runs the same
            // arithmetic in every lane, without masking.  HARVEST: a lane whose ray the leaf's box test kept out only collects light
            // distances, so it is struck from the result of every pair without a light's primitive (a scalar mask operation).
            float upper = mine ? h.t : -__builtin_inff();
            asm("" : "+v"(upper));  // (keeps the choice out of the loop: the compiler would redo it after every box test)
            const unsigned long long lights_only_lanes = HARVEST ? RRT_BALLOT(mine && lights_only) : 0ull;
            const unsigned long long whole_query_lanes = ~lights_only_lanes;
            uint32_t k = 0;
            if ((FEAT & (F_SPHERE | F_MEDIUM)) && (leaf_word & kLeafDirect) != 0u) {
                // The host marks a leaf whose only sphere or medium stands alone among triangles, and puts that record first.  One of a
                // kind cannot fill a dense group, so each lane whose ray meets the box tests its own ray here, with the record in scalar
                // registers; the answer competes with the dense groups' by the same (distance, slot) key.
                const cv4p rec = cpairs + kPairQuads * leaf_first;
                const v4f bx = rec[6], bz = rec[7];
                const bool box_met = AABB_MODE == kAabbExhaustive ? upper != -__builtin_inff() : pair_box_overlap_ch(bx.xy, bx.zw, bz.xy, finv, noi, sx, sy, sz, t_min, upper);
                const bool pass = box_met && !(HARVEST && lights_only && __float_as_uint(bz.z) == 0u);
                if (pass) {
                    const v4f q4 = rec[4], q5 = rec[5];
                    Hit hh;
                    hh.slot = -1;
                    hh.t = t_max;
                    hh.u = hh.v = hh.det = 0.0f;
                    test_slot<FEAT>(sc, __float_as_uint(q5.x), r, t_min, key, depth, hh);
                    if (hh.slot >= 0) {
                        const uint32_t light_a = HARVEST ? RRT_PAIR_LIGHT(__float_as_uint(q4.z)) : 0u;
                        if (HARVEST && light_a != 0u) light_t[lane * kHarvestLights + light_a - 1u] = hh.t;  // Sphere::pdf, shape.rs:142
                        if (!(HARVEST && lights_only)) {
                            const unsigned long long cand = ((unsigned long long)ord_bits(hh.t) << 32) | (unsigned long long)(((uint32_t)hh.slot + 1u) << 1);
                            direct_best = cand < direct_best ? cand : direct_best;
                        }
                    }
                }
                probe.pair_tested();
                k = 1;
            }
            // (the loop runs on the pair's number and a pointer that moves from box to box: its scalar instructions
            // count as much as its vector ones - the scalar unit serves a SIMD once in four cycles, the rate at which a SIMD issues
            // wave-wide vector work)
            // `at` = the pair's number times the size of a pair record: the byte offset of its record, and at the same time the pair
            // field of a list entry (records are 128 bytes; entries hold the pair from bit 7 up when harvesting, from bit 6 otherwise)
            static_assert(kPairQuads * 16u == 128u, "a list entry's pair field is the record's byte offset");
            const uint32_t at_end = (leaf_first + leaf_pairs) * 128u;
            const char* const boxes = (const char*)cpairs + 6 * 16;
            // Two pairs per turn (the host pads a leaf's records to an even number with a box no ray meets): the two boxes are loaded
            // together and waited for once, and the turn's scalar work - the step, the room check - is shared by both.
            auto list_pair = [&](uint32_t at, v4f bx, v4f bz) {
                const bool box_met = AABB_MODE == kAabbExhaustive ? upper != -__builtin_inff() : pair_box_overlap_ch(bx.xy, bx.zw, bz.xy, finv, noi, sx, sy, sz, t_min, upper);
                unsigned long long bal = RRT_BALLOT(box_met);
                if (HARVEST) bal &= whole_query_lanes | (((unsigned long long)__float_as_uint(bz.w) << 32) | (unsigned long long)__float_as_uint(bz.z));
                if (AABB_MODE == kAabbExhaustive) bal &= RRT_BALLOT(__float_as_uint(bx.x) != kPadBoxBits);  // (this mode has no box test to keep the padding out)
                const bool pass = __builtin_amdgcn_inverse_ballot_w64(bal);
                if (pass) {  // (the prefix count starts from the list's length: the place in the list in two instructions)
                    const uint32_t place = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, n_items));
                    if (HARVEST) items[place] = at | (lights_only ? 64u : 0u) | lane;
                    else items[place] = (at >> 1) | lane;
                }
                n_items += (uint32_t)__popcll(bal);
                probe.pair_tested();
            };
            // (only in the variants without textures, bump maps and media: the others are several times the instruction cache, and
            // a second copy of the loop's body costs them more than the shared scalar work saves - Example_Spec 3 % slower)
            constexpr bool kTwoPerTurn = (FEAT & (F_TEX | F_BUMP | F_MEDIUM)) == 0u;
            for (uint32_t at = (leaf_first + k) * 128u; at < at_end; at += kTwoPerTurn ? 256u : 128u) {
                const cv4p box = (cv4p)(boxes + at);
                const v4f bx0 = box[0], bz0 = box[1];  // (h.x, c.x, h.y, c.y), (h.z, c.z, lane mask in two words: see kPairLitMask)
                list_pair(at, bx0, bz0);
                if (kTwoPerTurn) {
                    const v4f bx1 = box[kPairQuads], bz1 = box[kPairQuads + 1];
                    list_pair(at + 128u, bx1, bz1);
                }
                if (n_items > CAP - 128u) {
                    probe.flushed(n_items);
                    sift_flush<FEAT, HARVEST, CAP>(sc, r, t_min, t_max, key, depth, items, n_items, best);
                    probe.dense_done();
                    n_items = 0;
                }
            }
            todo &= ~RRT_BALLOT(mine);
            if (HARVEST) break;
        }
        probe.flushed(n_items);
        probe.phase_done();
        sift_flush<FEAT, HARVEST, CAP>(sc, r, t_min, t_max, key, depth, items, n_items, best);
        probe.dense_done();
        if (pending) {
            if (HARVEST)
                for (uint32_t i = 0; i < kHarvestLights; ++i) harvest[i] = light_t[lane * kHarvestLights + i];
            unsigned long long w = best[lane];
            if ((FEAT & (F_SPHERE | F_MEDIUM)) && direct_best < w) w = direct_best;
            const uint32_t low = (uint32_t)w;
            if (low != 0u) {
                h.t = ord_float((uint32_t)(w >> 32));
                h.slot = (int)((low >> 1) - 1u);
                h.det = (low & 1u) ? -1.0f : 1.0f;  // only the sign is used (by the hit properties)
            }
            pending = false;
            if (!HARVEST && sp != 0) {
                --sp;
                cur = stack[sp * stride];
                running = true;
            }
        }
        if (HARVEST) break;
    }
    probe.store(lane);
    if (WANT_UV && (FEAT & (F_TEX | F_BUMP)) && h.slot >= 0) {
        // Triangle::get_hit_properties recomputes the barycentrics from the ray (shape.rs:384-400); so does this path
        // (callers that shade later, from a stored hit, pass WANT_UV = false and do this themselves)
        const cv4p cslots = sc.c_slots();
        const v4f q0 = cslots[3 * h.slot], q1 = cslots[3 * h.slot + 1], q2 = cslots[3 * h.slot + 2];
        if (RRT_INFO_KIND(__float_as_uint(q1.w)) == RRT_KIND_TRIANGLE) triangle_uvdet(xyz(q0), xyz(q1), xyz(q2), r, &h.u, &h.v, &h.det);
    }
    return h;
}

}  // namespace rrt
