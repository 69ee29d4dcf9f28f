/*
 * ort_tree.cpp -- builds the GPU acceleration structure on the host.
 *
 * Replaces the loose-octree construction of the reference (code/macos_main.mm:418-545
 * driving code/ray.cpp:1799-2045).  raycast_bvh (ray.cpp:624-822) returns the minimum-t
 * hit over all shapes whose node chain passes a conservative test, so ANY conservative
 * hierarchy yields the same (t, normal, material) up to bit-equal-t ties.
 * This one is laid out for divergent per-lane fetches on gfx950: 64-byte nodes that
 * carry both children's boxes, primitives reordered so that every leaf is a contiguous
 * run of 16-byte-aligned records of one kind, split by binned SAH.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <float.h>

#include "ort_scene.h"

namespace ort {

namespace {
constexpr float kPrologueBudget = 14.0f; /* analytic shapes worth up to this many box tests are tested outright (see build_tree); 10 -> 14 in round 2: the three
                                            large spheres of c2_analytic and testscene join their boxes, +5 % on both; the small scenes of the reference sit at 8.5 */

struct Box3 {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
    }
    void grow(const Box3 &b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    void grow_point(const float p[3]) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    uint32_t kind, index; /* index into the scene's source arrays (triangles: global id) */
    Box3 box;
    float centroid[3];
};

struct TriRef { uint32_t mesh, k; };

/* the reference's vector expressions, kept op-for-op so precomputed values carry the same bits */
inline void v_sub(const float a[3], const float b[3], float r[3]) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
inline void v_cross(const float a[3], const float b[3], float r[3]) { /* math.h:280-290 */
    r[0] = a[1] * b[2] - b[1] * a[2];
    r[1] = b[0] * a[2] - a[0] * b[2];
    r[2] = a[0] * b[1] - b[0] * a[1];
}
inline float v_len(const float a[3]) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); } /* math.h:173-177 */
inline bool near0(float v) { return v >= -0.000001f && v < 0.000001f; }
inline bool v_is0(const float a[3]) { return near0(a[0]) && near0(a[1]) && near0(a[2]); }       /* math.h:331-345 */
inline void v_normalize(const float a[3], float r[3]) {                                         /* math.h:298-310 */
    float l = v_len(a);
    float diff = l - 0.0f;
    if (!(diff >= -0.000001f && diff < 0.000001f)) { r[0] = a[0] / l; r[1] = a[1] / l; r[2] = a[2] / l; }
    else { r[0] = r[1] = r[2] = 0.0f; }
}

/* rotation_matrix_along_z (ray.cpp:8-33), rows b, c, a */
void cylinder_frame(const ort_cylinder &c, float rot[9], float *len) {
    const float z[3] = {0, 0, 1}, x[3] = {1, 0, 0};
    float axis[3] = {c.axis.x, c.axis.y, c.axis.z};
    float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(rot, id, sizeof(id));
    float cr[3];
    v_cross(axis, z, cr);
    if (!v_is0(cr)) {
        float a[3], b[3], t[3], cc[3];
        v_normalize(axis, a);
        v_cross(z, a, t);
        v_normalize(t, b);
        if (v_is0(b)) { v_cross(x, a, t); v_normalize(t, b); }
        v_cross(a, b, cc);
        memcpy(rot + 0, b, 12); memcpy(rot + 3, cc, 12); memcpy(rot + 6, a, 12);
    }
    *len = v_len(axis); /* ray.cpp:302 */
}

/* pad a primitive's box by a few ulps of its coordinates so the node test stays conservative */
void pad_box(Box3 *b) {
    for (int k = 0; k < 3; ++k) {
        float m = std::max(fabsf(b->lo[k]), fabsf(b->hi[k]));
        float pad = 4e-7f * m + 1e-7f;
        b->lo[k] -= pad;
        b->hi[k] += pad;
    }
}

struct Builder {
    Scene *scene;
    std::vector<Prim> prims;
    std::vector<TriRef> tri_refs;
    Tree *tree;
    static constexpr int kBins = 16;
    uint32_t kMaxLeafTri = 4;   /* tuning knobs (env ORT_LEAF_TRI / ORT_LEAF_OTHER); results do not depend on them */
    uint32_t kMaxLeafOther = 2;
    static constexpr uint32_t kDepthBudget = kTreeDepthBudget; /* ort_scene.h: the kernels' smallest traversal stack */

    static constexpr float kNodeCost = 1.0f, kPrimCost = 1.2f;
    double sah_sum = 0;
    float root_area = 1;

    static uint32_t log2_ceil(uint32_t n) {
        uint32_t l = 0;
        while ((1u << l) < n) ++l;
        return l;
    }

    uint32_t emit_leaf(uint32_t begin, uint32_t end) {
        uint32_t kind = prims[begin].kind, count = end - begin, first = 0;
        switch (kind) {
        case PRIM_TRI:
            first = (uint32_t)tree->tris.size();
            for (uint32_t i = begin; i < end; ++i) {
                const TriRef &tr = tri_refs[prims[i].index];
                const HostMesh &m = scene->meshes[tr.mesh];
                const float *v0 = &m.vertices[3 * (size_t)m.indices[tr.k]];
                const float *v1 = &m.vertices[3 * (size_t)m.indices[tr.k + 1]];
                const float *v2 = &m.vertices[3 * (size_t)m.indices[tr.k + 2]];
                DevTri t;
                memcpy(t.v0, v0, 12);
                v_sub(v1, v0, t.e1); /* ray.cpp:87 */
                v_sub(v2, v0, t.e2); /* ray.cpp:88 */
                v_cross(t.e1, t.e2, t.n); /* ray.cpp:110 */
                tree->tri_slot[prims[i].index] = (uint32_t)tree->tris.size();
                tree->tris.push_back(t);
                tree->tri_mat.push_back(m.mat);
            }
            break;
        case PRIM_SPHERE:
            first = (uint32_t)tree->spheres.size();
            for (uint32_t i = begin; i < end; ++i) {
                const ort_sphere &s = scene->spheres[prims[i].index];
                tree->sphere_slot[prims[i].index] = (uint32_t)tree->spheres.size();
                tree->spheres.push_back(DevSphere{{s.center.x, s.center.y, s.center.z}, s.r});
                tree->sphere_mat.push_back(s.mat);
            }
            break;
        case PRIM_BOX:
            first = (uint32_t)tree->boxes.size();
            for (uint32_t i = begin; i < end; ++i) {
                const ort_box &b = scene->boxes[prims[i].index];
                tree->box_slot[prims[i].index] = (uint32_t)tree->boxes.size();
                tree->boxes.push_back(DevBox{{b.min.x, b.min.y, b.min.z}, 0, {b.max.x, b.max.y, b.max.z}, 0});
                tree->box_mat.push_back(b.mat);
            }
            break;
        default:
            first = (uint32_t)tree->cyls.size();
            for (uint32_t i = begin; i < end; ++i) {
                const ort_cylinder &c = scene->cylinders[prims[i].index];
                DevCyl d{};
                d.base[0] = c.base.x; d.base[1] = c.base.y; d.base[2] = c.base.z;
                d.r = c.r;
                cylinder_frame(c, d.rot, &d.len);
                tree->cyl_slot[prims[i].index] = (uint32_t)tree->cyls.size();
                tree->cyls.push_back(d);
                tree->cyl_mat.push_back(c.mat);
            }
            break;
        }
        tree->leaf_count++;
        tree->max_leaf_prims = std::max(tree->max_leaf_prims, count);
        return make_leaf(kind, first, count);
    }

    Box3 bounds(uint32_t begin, uint32_t end) const {
        Box3 b;
        b.reset();
        for (uint32_t i = begin; i < end; ++i) b.grow(prims[i].box);
        return b;
    }

    bool homogeneous(uint32_t begin, uint32_t end) const {
        for (uint32_t i = begin + 1; i < end; ++i)
            if (prims[i].kind != prims[begin].kind) return false;
        return true;
    }

    /* returns the child word for prims [begin,end); box = their bounds */
    uint32_t build(uint32_t begin, uint32_t end, const Box3 &box, uint32_t depth) {
        tree->max_depth = std::max(tree->max_depth, depth);
        uint32_t count = end - begin;
        bool same_kind = homogeneous(begin, end);
        uint32_t max_leaf = (prims[begin].kind == PRIM_TRI) ? kMaxLeafTri : kMaxLeafOther;
        bool out_of_depth = depth >= kDepthBudget;
        if (same_kind && (count == 1 || (out_of_depth && count <= MAX_LEAF_PRIMS))) {
            sah_sum += (double)kPrimCost * count * box.half_area();
            return emit_leaf(begin, end);
        }

        uint32_t mid = begin;
        bool have_split = false;
        bool force_median = (depth + log2_ceil(count) + 2 >= kDepthBudget);
        if (!same_kind && count <= max_leaf) {
            /* small mixed group: separate the first kind from the rest */
            uint32_t k0 = prims[begin].kind;
            mid = (uint32_t)(std::partition(prims.begin() + begin, prims.begin() + end,
                                            [&](const Prim &p) { return p.kind == k0; }) - prims.begin());
            have_split = (mid > begin && mid < end);
        }
        if (!have_split && !force_median) {
            Box3 cb;
            cb.reset();
            for (uint32_t i = begin; i < end; ++i) cb.grow_point(prims[i].centroid);
            float best_cost = FLT_MAX;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                float extent = cb.hi[axis] - cb.lo[axis];
                if (!(extent > 0)) continue;
                Box3 bin_box[kBins];
                uint32_t bin_n[kBins];
                for (int b = 0; b < kBins; ++b) { bin_box[b].reset(); bin_n[b] = 0; }
                float scale = kBins / extent;
                for (uint32_t i = begin; i < end; ++i) {
                    int b = (int)((prims[i].centroid[axis] - cb.lo[axis]) * scale);
                    b = std::max(0, std::min(kBins - 1, b));
                    bin_box[b].grow(prims[i].box);
                    bin_n[b]++;
                }
                float right_area[kBins];
                uint32_t right_n[kBins];
                Box3 acc;
                acc.reset();
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    n += bin_n[b];
                    right_area[b] = acc.half_area();
                    right_n[b] = n;
                }
                acc.reset();
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bin_box[b]);
                    n += bin_n[b];
                    if (n == 0 || right_n[b + 1] == 0) continue;
                    float cost = acc.half_area() * n + right_area[b + 1] * right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                float area = std::max(box.half_area(), 1e-30f);
                float split_cost = kNodeCost + kPrimCost * best_cost / area;
                float leaf_cost = kPrimCost * count;
                if (same_kind && count <= max_leaf && leaf_cost <= split_cost) {
                    sah_sum += (double)kPrimCost * count * box.half_area();
                    return emit_leaf(begin, end);
                }
                float lo = cb.lo[best_axis], scale = kBins / (cb.hi[best_axis] - cb.lo[best_axis]);
                mid = (uint32_t)(std::partition(prims.begin() + begin, prims.begin() + end, [&](const Prim &p) {
                          int b = (int)((p.centroid[best_axis] - lo) * scale);
                          b = std::max(0, std::min(kBins - 1, b));
                          return b <= best_bin;
                      }) - prims.begin());
                have_split = (mid > begin && mid < end);
            } else if (same_kind && count <= max_leaf) {
                sah_sum += (double)kPrimCost * count * box.half_area();
                return emit_leaf(begin, end);
            }
        }
        if (!have_split) {
            /* object median along the widest box axis (also the balanced fallback near the depth budget) */
            int axis = 0;
            float ext[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            mid = begin + count / 2;
            std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                             [axis](const Prim &a, const Prim &b) { return a.centroid[axis] < b.centroid[axis]; });
        }

        uint32_t node_index = (uint32_t)tree->nodes.size();
        tree->nodes.push_back(DevNode{});
        sah_sum += (double)kNodeCost * box.half_area();
        Box3 lb = bounds(begin, mid), rb = bounds(mid, end);
        uint32_t c0 = build(begin, mid, lb, depth + 1);
        uint32_t c1 = build(mid, end, rb, depth + 1);
        DevNode &n = tree->nodes[node_index];
        memcpy(n.lo0, lb.lo, 12); memcpy(n.hi0, lb.hi, 12);
        memcpy(n.lo1, rb.lo, 12); memcpy(n.hi1, rb.hi, 12);
        n.child0 = c0; n.child1 = c1;
        return node_index | ((c0 | c1) & SPHERE_BELOW_BIT);
    }
};

} // namespace

DevMaterial make_dev_material(const ort_material &m) {
    DevMaterial d{};
    d.diffuse[0] = m.diffuse.x; d.diffuse[1] = m.diffuse.y; d.diffuse[2] = m.diffuse.z; d.ior = m.ior;
    d.specular[0] = m.specular[0]; d.specular[1] = m.specular[1]; d.specular[2] = m.specular[2];
    d.is_light = m.is_light ? 1u : 0u;
    d.transmission[0] = m.transmission.x; d.transmission[1] = m.transmission.y; d.transmission[2] = m.transmission.z;
    d.emit[0] = m.emit.x; d.emit[1] = m.emit.y; d.emit[2] = m.emit.z;
    /* ray.cpp:1010-1018: lengths, their sum, the three ratios (0/0 = NaN for an all-zero material, sic) */
    float kd = v_len(d.diffuse), ks = v_len(d.specular), kt = v_len(d.transmission);
    float s = kd + ks + kt;
    d.pd_c = kd / s; d.ps_c = ks / s; d.pt_c = kt / s;
    /* ray.cpp:939: Ed = Kd / pi_32, three divides */
    const float pi_32 = 3.14159265358979323846264338327950288419716939937510582097494459230f;
    d.ed[0] = d.diffuse[0] / pi_32; d.ed[1] = d.diffuse[1] / pi_32; d.ed[2] = d.diffuse[2] / pi_32;
    return d;
}

void cylinder_frame_for(const ort_cylinder &c, float rot[9], float *len) { cylinder_frame(c, rot, len); }

/* The first kTreeletNodes interior nodes in breadth-first order get the indices [0, kTreeletNodes): the kernel keeps
   that top of the tree in LDS (ort_lane.h, kTabTreelet), where most node visits happen.  Pure renumbering. */
static void renumber_top_levels(Tree *tree, uint32_t top) {
    const size_t n = tree->nodes.size();
    if (n <= 2) return;
    std::vector<uint32_t> order; /* new index -> old index */
    std::vector<uint8_t> taken(n, 0);
    order.push_back(0);
    taken[0] = 1;
    for (size_t h = 0; h < order.size() && order.size() < top; ++h) {
        const DevNode &nd = tree->nodes[order[h]];
        const uint32_t ch[2] = {nd.child0, nd.child1};
        for (uint32_t c : ch) {
            if (c & LEAF_BIT) continue;
            const uint32_t i = c & NODE_INDEX_MASK;
            if (i < n && !taken[i] && order.size() < top) { taken[i] = 1; order.push_back(i); }
        }
    }
    for (uint32_t i = 0; i < n; ++i)
        if (!taken[i]) order.push_back(i);
    std::vector<uint32_t> new_of(n);
    for (uint32_t k = 0; k < n; ++k) new_of[order[k]] = k;
    std::vector<DevNode> out(n);
    for (uint32_t k = 0; k < n; ++k) {
        DevNode nd = tree->nodes[order[k]];
        if (!(nd.child0 & LEAF_BIT)) nd.child0 = (nd.child0 & ~NODE_INDEX_MASK) | new_of[nd.child0 & NODE_INDEX_MASK];
        if (!(nd.child1 & LEAF_BIT)) nd.child1 = (nd.child1 & ~NODE_INDEX_MASK) | new_of[nd.child1 & NODE_INDEX_MASK];
        out[k] = nd;
    }
    tree->nodes.swap(out);
}

/* The 4-wide form of the finished binary tree: a wide node adopts its binary node's two children and then, while it has
   fewer than four, replaces the interior child with the largest box by that child's own two children (the usual
   surface-area-greedy collapse).  Leaves and primitive arrays are shared with the binary tree.  Nodes are emitted
   breadth-first, so the top of the tree has the lowest indices (the kernel keeps the first kTreeletNodes / 2 in LDS). */
static void collapse_to_wide(Tree *tree) {
    tree->nodes4.clear();
    tree->max_depth4 = 0;
    const std::vector<DevNode> &bn = tree->nodes;
    if (bn.empty()) return;
    struct Cand { uint32_t word; float lo[3], hi[3]; };
    auto area = [](const Cand &c) {
        const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : dx * dy + dy * dz + dz * dx;
    };
    auto children_of = [&bn](uint32_t bi, Cand out[2]) {
        const DevNode &n = bn[bi];
        out[0].word = n.child0; memcpy(out[0].lo, n.lo0, 12); memcpy(out[0].hi, n.hi0, 12);
        out[1].word = n.child1; memcpy(out[1].lo, n.lo1, 12); memcpy(out[1].hi, n.hi1, 12);
    };
    struct Todo { uint32_t binary, wide, depth; };
    std::vector<Todo> queue;
    tree->nodes4.push_back(DevNode4{});
    queue.push_back(Todo{0u, 0u, 1u});
    for (size_t h = 0; h < queue.size(); ++h) {
        const Todo td = queue[h];
        tree->max_depth4 = std::max(tree->max_depth4, td.depth);
        Cand c[4];
        uint32_t n = 0;
        {
            Cand two[2];
            children_of(td.binary, two);
            for (int k = 0; k < 2; ++k)
                if (two[k].word != EMPTY_CHILD) c[n++] = two[k];
        }
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (uint32_t k = 0; k < n; ++k)
                if (!(c[k].word & LEAF_BIT) && area(c[k]) > best) { best = area(c[k]); pick = (int)k; }
            if (pick < 0) break;
            Cand two[2];
            children_of(c[pick].word & NODE_INDEX_MASK, two);
            uint32_t live = 0;
            for (int k = 0; k < 2; ++k)
                if (two[k].word != EMPTY_CHILD) live++;
            if (n - 1 + live > 4) break;
            /* the adopted children take the expanded child's place, in their own order */
            Cand next[4];
            uint32_t m = 0;
            for (uint32_t k = 0; k < n; ++k) {
                if ((int)k != pick) { next[m++] = c[k]; continue; }
                for (int j = 0; j < 2; ++j)
                    if (two[j].word != EMPTY_CHILD) next[m++] = two[j];
            }
            n = m;
            for (uint32_t k = 0; k < n; ++k) c[k] = next[k];
        }
        DevNode4 w{};
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < n) {
                w.lox[k] = c[k].lo[0]; w.loy[k] = c[k].lo[1]; w.loz[k] = c[k].lo[2];
                w.hix[k] = c[k].hi[0]; w.hiy[k] = c[k].hi[1]; w.hiz[k] = c[k].hi[2];
                if (c[k].word & LEAF_BIT) {
                    w.child[k] = c[k].word;
                } else {
                    const uint32_t wi = (uint32_t)tree->nodes4.size();
                    tree->nodes4.push_back(DevNode4{});
                    queue.push_back(Todo{c[k].word & NODE_INDEX_MASK, wi, td.depth + 1u});
                    w.child[k] = wi | (c[k].word & SPHERE_BELOW_BIT);
                }
            } else {
                w.lox[k] = w.loy[k] = w.loz[k] = FLT_MAX;
                w.hix[k] = w.hiy[k] = w.hiz[k] = -FLT_MAX;
                w.child[k] = EMPTY_CHILD;
            }
        }
        tree->nodes4[td.wide] = w;
    }
}

int build_tree(Scene *scene, std::string *err) {
    Tree fresh;
    scene->tree = fresh;
    Tree *tree = &scene->tree;
    Builder b;
    b.scene = scene;
    b.tree = tree;

    for (uint32_t mi = 0; mi < scene->meshes.size(); ++mi) {
        const HostMesh &m = scene->meshes[mi];
        for (uint32_t k = 0; k + 2 < m.indices.size(); k += 3) {
            Prim p;
            p.kind = PRIM_TRI;
            p.index = (uint32_t)b.tri_refs.size();
            b.tri_refs.push_back(TriRef{mi, k});
            p.box.reset();
            for (int c = 0; c < 3; ++c) p.box.grow_point(&m.vertices[3 * (size_t)m.indices[k + c]]);
            b.prims.push_back(p);
        }
    }
    for (uint32_t i = 0; i < scene->spheres.size(); ++i) {
        const ort_sphere &s = scene->spheres[i];
        Prim p;
        p.kind = PRIM_SPHERE; p.index = i;
        /* every ray the intersector can report a hit for passes within sqrt(r^2 + 1e-5) of the
           centre (tangent branch, ray.cpp:145,174) */
        float r = sqrtf(s.r * s.r + 0.00001f) * 1.00001f + 1e-6f;
        p.box.lo[0] = s.center.x - r; p.box.lo[1] = s.center.y - r; p.box.lo[2] = s.center.z - r;
        p.box.hi[0] = s.center.x + r; p.box.hi[1] = s.center.y + r; p.box.hi[2] = s.center.z + r;
        b.prims.push_back(p);
    }
    for (uint32_t i = 0; i < scene->boxes.size(); ++i) {
        const ort_box &bx = scene->boxes[i];
        Prim p;
        p.kind = PRIM_BOX; p.index = i;
        p.box.lo[0] = std::min(bx.min.x, bx.max.x); p.box.lo[1] = std::min(bx.min.y, bx.max.y); p.box.lo[2] = std::min(bx.min.z, bx.max.z);
        p.box.hi[0] = std::max(bx.min.x, bx.max.x); p.box.hi[1] = std::max(bx.min.y, bx.max.y); p.box.hi[2] = std::max(bx.min.z, bx.max.z);
        b.prims.push_back(p);
    }
    for (uint32_t i = 0; i < scene->cylinders.size(); ++i) {
        const ort_cylinder &c = scene->cylinders[i];
        Prim p;
        p.kind = PRIM_CYL; p.index = i;
        /* bound what the intersector actually tests: the segment base .. base + len * a in the
           frame of ray.cpp:8-33 (a = +z whenever axis x z ~ 0, even for axis = -z), radius r */
        float rot[9], len;
        cylinder_frame(c, rot, &len);
        const float *a = rot + 6;
        float p0[3] = {c.base.x, c.base.y, c.base.z};
        float p1[3] = {p0[0] + len * a[0], p0[1] + len * a[1], p0[2] + len * a[2]};
        p.box.reset();
        p.box.grow_point(p0);
        p.box.grow_point(p1);
        float r = fabsf(c.r) * 1.0001f;
        for (int k = 0; k < 3; ++k) {
            float e = r * sqrtf(std::max(0.0f, 1.0f - a[k] * a[k])) + 1e-5f * r;
            p.box.lo[k] -= e;
            p.box.hi[k] += e;
        }
        b.prims.push_back(p);
    }
    /* The reference's sphere and cylinder intersectors solve their quadratics in f32: the discriminant
       b^2 - a c cancels, and what they report is the exact hit on a shape whose squared radius is off by up to a
       few ulps of |origin - shape|^2 -- at 21 units from an r = 0.05 cylinder a "hit" 2.8e-4 outside it was seen
       (tools/stress_parity.py, testscene, seed 150229).  The fast tree's boxes must contain every hit the
       intersector can REPORT, so quadric boxes grow to the radius sqrt(r^2 + 16 * 2^-23 * D^2), D = the farthest a ray
       origin can be (diagonal of everything in the scene and the camera), plus the same bound linearly for the
       cap planes. */
    {
        Box3 all;
        all.reset();
        for (const Prim &p : b.prims) all.grow(p.box);
        const float cam[3] = {scene->camera_p.x, scene->camera_p.y, scene->camera_p.z};
        all.grow_point(cam);
        float d2 = 0;
        for (int k = 0; k < 3; ++k) { float e = all.hi[k] - all.lo[k] + 0.5f; d2 += e * e; }
        const float slack2 = 16.0f * 1.1920929e-7f * d2, slack1 = 16.0f * 1.1920929e-7f * sqrtf(d2);
        for (Prim &p : b.prims) {
            if (p.kind == PRIM_SPHERE) {
                const ort_sphere &sp = scene->spheres[p.index];
                const float r0 = sqrtf(sp.r * sp.r + 0.00001f), r1 = sqrtf(sp.r * sp.r + 0.00001f + slack2);
                for (int k = 0; k < 3; ++k) { p.box.lo[k] -= (r1 - r0) * 1.0001f + slack1; p.box.hi[k] += (r1 - r0) * 1.0001f + slack1; }
            } else if (p.kind == PRIM_CYL) {
                const ort_cylinder &c = scene->cylinders[p.index];
                const float r0 = fabsf(c.r), r1 = sqrtf(c.r * c.r + slack2);
                for (int k = 0; k < 3; ++k) { p.box.lo[k] -= (r1 - r0) * 1.0002f + slack1; p.box.hi[k] += (r1 - r0) * 1.0002f + slack1; }
            }
        }
    }
    for (Prim &p : b.prims) {
        pad_box(&p.box);
        for (int k = 0; k < 3; ++k) {
            if (!(p.box.lo[k] == p.box.lo[k]) || !(p.box.hi[k] == p.box.hi[k]) || fabsf(p.box.lo[k]) > 1e30f || fabsf(p.box.hi[k]) > 1e30f) {
                *err = "scene contains a non-finite or oversized primitive";
                return ORT_ERR_INVALID;
            }
            p.centroid[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        }
    }
    if (b.prims.size() > 0x00ffffffu) { *err = "more than 2^24 primitives"; return ORT_ERR_UNSUPPORTED; }
    tree->tri_slot.assign(b.tri_refs.size(), 0);
    tree->sphere_slot.assign(scene->spheres.size(), 0);
    tree->box_slot.assign(scene->boxes.size(), 0);
    tree->cyl_slot.assign(scene->cylinders.size(), 0);

    /* node 0 always exists and is interior, so traversal starts uniformly */
    tree->nodes.reserve(b.prims.size() + 2);
    if (b.prims.empty()) {
        DevNode n{};
        for (int k = 0; k < 3; ++k) { n.lo0[k] = n.lo1[k] = FLT_MAX; n.hi0[k] = n.hi1[k] = -FLT_MAX; }
        n.child0 = n.child1 = EMPTY_CHILD;
        tree->nodes.push_back(n);
    } else {
        Box3 all = b.bounds(0, (uint32_t)b.prims.size());
        b.root_area = std::max(all.half_area(), 1e-30f);
        if (const char *e = getenv("ORT_LEAF_TRI")) b.kMaxLeafTri = std::max(1, std::min((int)MAX_LEAF_PRIMS, atoi(e)));
        if (const char *e = getenv("ORT_LEAF_OTHER")) b.kMaxLeafOther = std::max(1, std::min((int)MAX_LEAF_PRIMS, atoi(e)));
        /* the few large analytic shapes (room boxes, lights) and the meshes get separate subtrees under
           the root: the analytic side is visited first and usually fixes best_t before any mesh node */
        const char *split_env = getenv("ORT_TREE_SPLIT_KINDS");
        bool split_kinds = split_env ? atoi(split_env) != 0 : true;
        uint32_t n_analytic = (uint32_t)(std::partition(b.prims.begin(), b.prims.end(), [](const Prim &p) { return p.kind != PRIM_TRI; }) - b.prims.begin());
        /* A handful of analytic shapes (the room, the lights) is cheaper to test outright, all lanes of a
           wave on the same shape at the same time, than to reach through tree nodes in divergent lanes:
           they stay out of the tree and form the ray's prologue (tuned on MI355X, profiles/r01_tuning.md) */
        const char *pro_env = getenv("ORT_ANALYTIC_PROLOGUE");
        /* (trees of every size: the collapse rounds 1 and 2 saw with it on the 1M-triangle scene was the lock pool of
           the exact fallback, ort_lane.h recast_exactly -- with that cured the prologue is worth +23 % there) */
        const float pro_budget = pro_env ? (float)atof(pro_env) : kPrologueBudget;
        /* cheapest kinds first (a sphere test costs about 1.5 box tests, a cylinder about 5), as many as fit */
        auto cost_of = [](const Prim &p) { return p.kind == PRIM_BOX ? 1.0f : p.kind == PRIM_SPHERE ? 1.5f : 5.0f; };
        /* within a kind the largest shapes first: they are the ones most rays hit */
        std::stable_sort(b.prims.begin(), b.prims.begin() + n_analytic, [&](const Prim &x, const Prim &y) {
            return cost_of(x) != cost_of(y) ? cost_of(x) < cost_of(y) : x.box.half_area() > y.box.half_area();
        });
        uint32_t n_pro = 0;
        for (float spent = 0; n_pro < n_analytic && spent + cost_of(b.prims[n_pro]) <= pro_budget; ++n_pro) spent += cost_of(b.prims[n_pro]);
        tree->analytic_prologue = n_pro > 0;
        tree->pro_boxes = tree->pro_spheres = tree->pro_cyls = 0;
        for (uint32_t i = 0; i < n_pro; ++i) { /* into the shape arrays (slots 0.. of each kind), not the tree */
            (void)b.emit_leaf(i, i + 1);
            (b.prims[i].kind == PRIM_BOX ? tree->pro_boxes : b.prims[i].kind == PRIM_SPHERE ? tree->pro_spheres : tree->pro_cyls)++;
        }
        tree->leaf_count = 0;
        const uint32_t n_all = (uint32_t)b.prims.size();
        uint32_t root;
        if (n_pro == n_all) { /* nothing left for the tree */
            DevNode n{};
            for (int k = 0; k < 3; ++k) { n.lo0[k] = n.lo1[k] = FLT_MAX; n.hi0[k] = n.hi1[k] = -FLT_MAX; }
            n.child0 = n.child1 = EMPTY_CHILD;
            tree->nodes.push_back(n);
            tree->sah_cost = 0;
            tree->built = true;
            return ORT_OK;
        }
        all = b.bounds(n_pro, n_all);
        if (split_kinds && n_analytic > n_pro && n_analytic < n_all) {
            tree->nodes.push_back(DevNode{});
            Box3 ab = b.bounds(n_pro, n_analytic), tb = b.bounds(n_analytic, n_all);
            uint32_t c0 = b.build(n_pro, n_analytic, ab, 1);
            uint32_t c1 = b.build(n_analytic, n_all, tb, 1);
            DevNode &n = tree->nodes[0];
            memcpy(n.lo0, ab.lo, 12); memcpy(n.hi0, ab.hi, 12);
            memcpy(n.lo1, tb.lo, 12); memcpy(n.hi1, tb.hi, 12);
            n.child0 = c0; n.child1 = c1;
            root = (c0 | c1) & SPHERE_BELOW_BIT;
        } else {
            root = b.build(n_pro, n_all, all, 0);
        }
        if (!(root & LEAF_BIT) && (root & NODE_INDEX_MASK) != 0) { *err = "internal: root is not node 0"; return ORT_ERR_INVALID; }
        if (root & LEAF_BIT) {
            DevNode n{};
            memcpy(n.lo0, all.lo, 12); memcpy(n.hi0, all.hi, 12);
            for (int k = 0; k < 3; ++k) { n.lo1[k] = FLT_MAX; n.hi1[k] = -FLT_MAX; }
            n.child0 = root; n.child1 = EMPTY_CHILD;
            tree->nodes.push_back(n);
        }
        tree->sah_cost = (float)(b.sah_sum / b.root_area);
    }
    if (tree->max_depth > kTreeDepthBudget) { /* cannot happen (the builder forces median splits near the budget): a tripwire for builder changes */
        *err = "internal: fast tree deeper than the traversal stacks";
        return ORT_ERR_UNSUPPORTED;
    }
    renumber_top_levels(tree, kTreeletNodes);
    collapse_to_wide(tree);
    tree->built = true;
    return ORT_OK;
}

} // namespace ort
