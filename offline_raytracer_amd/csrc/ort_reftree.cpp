/*
 * ort_reftree.cpp -- the reference-compatible loose octree (host build).
 *
 * Why the product needs it although traversal runs on its own tree (ort_tree.cpp):
 * raycast_bvh (code/ray.cpp:776-812) enqueues a child only if the ray origin is inside
 * the child's AABB (half-open, math.h:1156-1169) or the slab test enters it at
 * t >= 1e-6.  A bounce origin that lies ON a node face -- routine for grazing rays on
 * the axis-aligned room boxes, whose faces are node faces -- therefore hides everything
 * in that node from the reference.  That is part of the reference's output, and it
 * depends on the reference's node boxes.  So this file rebuilds exactly those boxes
 * (same f32 arithmetic as ray.cpp:1476-1522,1675-1777,1799-1948 driven in the order of
 * macos_main.mm:418-538, including main()'s inert CSG shape) and flattens, for every
 * primitive, the chain of node boxes above it.  The kernel checks the winner of its own
 * traversal against that chain and falls back to walking this octree when the reference
 * would not have seen it ("Exactness").
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <float.h>

#include "ort_scene.h"

namespace ort {

namespace {

struct F3 { float x, y, z; };
inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
/* types.h:50-51 macro semantics */
inline float lo2(float a, float b) { return (a < b) ? a : b; }
inline float hi2(float a, float b) { return (a > b) ? a : b; }
inline F3 vmin(F3 a, F3 b) { return f3(lo2(a.x, b.x), lo2(a.y, b.y), lo2(a.z, b.z)); }
inline F3 vmax(F3 a, F3 b) { return f3(hi2(a.x, b.x), hi2(a.y, b.y), hi2(a.z, b.z)); }

struct Extent { F3 center, half; };
inline Extent from_min_max(F3 lo, F3 hi) { /* ray.cpp:1706-1707 and siblings */
    Extent e;
    e.center = 0.5f * (lo + hi);
    e.half = hi - e.center;
    return e;
}

enum : uint8_t { S_SPHERE = 1, S_CYL = 2, S_BOX = 3, S_MESH = 4, S_TRI = 5, S_CSG = 6 }; /* ray.h:97-106 */

struct Shape { uint8_t type; uint32_t index; };

struct Node {
    int32_t first_child = -1;
    int32_t parent = -1;
    bool is_leaf = false;
    std::vector<Shape> shapes;
    F3 lo{FLT_MAX, FLT_MAX, FLT_MAX};
    F3 hi{FLT_MIN, FLT_MIN, FLT_MIN}; /* sic: smallest positive float (ray.cpp:1761, macos_main.mm:424) */
};

struct TriRef { uint32_t mesh, k; };

struct RefBuilder {
    const Scene *scene;
    std::vector<Node> nodes;
    std::vector<TriRef> tris;
    F3 csg_lo, csg_hi;
    uint32_t depth_limit = 10; /* macos_main.mm:474 */

    F3 vertex(uint32_t mesh, uint32_t i) const {
        const float *p = &scene->meshes[mesh].vertices[3 * (size_t)i];
        return f3(p[0], p[1], p[2]);
    }

    Extent extent(Shape s) const { /* ray.cpp:1675-1746 */
        switch (s.type) {
        case S_SPHERE: {
            const ort_sphere &sp = scene->spheres[s.index];
            Extent e;
            e.center = f3(sp.center.x, sp.center.y, sp.center.z);
            e.half = sp.r * f3(1, 1, 1);
            return e;
        }
        case S_CYL: {
            const ort_cylinder &c = scene->cylinders[s.index];
            F3 base = f3(c.base.x, c.base.y, c.base.z), axis = f3(c.axis.x, c.axis.y, c.axis.z);
            F3 other = base + axis;
            float aa = axis.x * axis.x + axis.y * axis.y + axis.z * axis.z;
            F3 q = f3((axis.x * axis.x) / aa, (axis.y * axis.y) / aa, (axis.z * axis.z) / aa);
            F3 e = c.r * (f3(1, 1, 1) - f3(sqrtf(q.x), sqrtf(q.y), sqrtf(q.z)));
            return from_min_max(vmin(base - e, other - e), vmax(base + e, other + e));
        }
        case S_BOX: {
            const ort_box &b = scene->boxes[s.index];
            return from_min_max(f3(b.min.x, b.min.y, b.min.z), f3(b.max.x, b.max.y, b.max.z));
        }
        case S_MESH: {
            const HostMesh &m = scene->meshes[s.index];
            return from_min_max(f3(m.aabb_min.x, m.aabb_min.y, m.aabb_min.z), f3(m.aabb_max.x, m.aabb_max.y, m.aabb_max.z));
        }
        case S_TRI: {
            const TriRef &t = tris[s.index];
            const HostMesh &m = scene->meshes[t.mesh];
            F3 a = vertex(t.mesh, m.indices[t.k]), b = vertex(t.mesh, m.indices[t.k + 1]), c = vertex(t.mesh, m.indices[t.k + 2]);
            return from_min_max(vmin(vmin(a, b), c), vmax(vmax(a, b), c));
        }
        default: return from_min_max(csg_lo, csg_hi);
        }
    }

    void grow(F3 *lo, F3 *hi, Shape s) const { /* ray.cpp:1765-1777 */
        Extent e = extent(s);
        *lo = vmin(*lo, e.center - e.half);
        *hi = vmax(*hi, e.center + e.half);
    }

    /* ray.cpp:1476-1522 + platform.h:140-162: slot bit0 = x >= c, bit1 = y >= c, bit2 = z >= c */
    static uint32_t child_of(F3 c, F3 half, F3 p, F3 *cc, F3 *ch) {
        uint32_t slot = 0;
        *ch = 0.5f * half;
        *cc = c;
        if (p.x >= c.x) { slot |= 1; cc->x += ch->x; } else { cc->x -= ch->x; }
        if (p.y >= c.y) { slot |= 2; cc->y += ch->y; } else { cc->y -= ch->y; }
        if (p.z >= c.z) { slot |= 4; cc->z += ch->z; } else { cc->z -= ch->z; }
        return slot;
    }

    void insert(uint32_t ni, F3 c, F3 half, uint32_t depth, Shape s) { /* ray.cpp:1799-1948 */
        grow(&nodes[ni].lo, &nodes[ni].hi, s);
        if (depth >= depth_limit) { nodes[ni].shapes.push_back(s); return; }
        if (nodes[ni].first_child >= 0) {
            F3 cc, ch;
            uint32_t slot = child_of(c, half, extent(s).center, &cc, &ch);
            insert((uint32_t)nodes[ni].first_child + slot, cc, ch, depth + 1, s);
            return;
        }
        if (nodes[ni].shapes.empty()) { nodes[ni].shapes.push_back(s); return; }
        /* occupied leaf: open eight children, re-insert residents, then the newcomer */
        uint32_t first = (uint32_t)nodes.size();
        for (int k = 0; k < 8; ++k) {
            Node ch;
            ch.is_leaf = true;
            ch.parent = (int32_t)ni;
            nodes.push_back(ch);
        }
        nodes[ni].first_child = (int32_t)first;
        std::vector<Shape> residents;
        residents.swap(nodes[ni].shapes);
        for (Shape r : residents) {
            F3 cc, ch;
            uint32_t slot = child_of(c, half, extent(r).center, &cc, &ch);
            insert(first + slot, cc, ch, depth + 1, r);
        }
        nodes[ni].is_leaf = false;
        F3 cc, ch;
        uint32_t slot = child_of(c, half, extent(s).center, &cc, &ch);
        insert(first + slot, cc, ch, depth + 1, s);
    }
};

} // namespace

int build_ref_tree(Scene *scene, std::string *err) {
    RefTree fresh;
    scene->ref = fresh;
    RefTree *out = &scene->ref;
    const Tree &tree = scene->tree;
    if (!tree.built) { *err = "internal: fast tree must be built before the reference octree"; return ORT_ERR_STATE; }

    RefBuilder b;
    b.scene = scene;
    for (uint32_t mi = 0; mi < scene->meshes.size(); ++mi)
        for (uint32_t k = 0; k + 2 < scene->meshes[mi].indices.size(); k += 3) b.tris.push_back(TriRef{mi, k});

    Node root; /* zero(top_most_node): is_leaf = false (macos_main.mm:421-424) */
    b.nodes.push_back(root);
    /* root AABB, in main()'s order (macos_main.mm:426-458) */
    for (uint32_t i = 0; i < scene->meshes.size(); ++i) b.grow(&b.nodes[0].lo, &b.nodes[0].hi, Shape{S_MESH, i});
    for (uint32_t i = 0; i < scene->cylinders.size(); ++i) b.grow(&b.nodes[0].lo, &b.nodes[0].hi, Shape{S_CYL, i});
    for (uint32_t i = 0; i < scene->boxes.size(); ++i) b.grow(&b.nodes[0].lo, &b.nodes[0].hi, Shape{S_BOX, i});
    for (uint32_t i = 0; i < scene->spheres.size(); ++i) b.grow(&b.nodes[0].lo, &b.nodes[0].hi, Shape{S_SPHERE, i});
    if (scene->reference_csg) {
        /* macos_main.mm:322-332,460-469: sphere r 0.35 and box +-0.3 around (0,0,0.8); only its AABB matters */
        F3 c = f3(0, 0, 0.8f);
        F3 lo = FLT_MAX * f3(1, 1, 1), hi = FLT_MIN * f3(1, 1, 1);
        F3 sh = 0.35f * f3(1, 1, 1);
        lo = vmin(lo, c - sh); hi = vmax(hi, c + sh);
        Extent bx = from_min_max(c - f3(0.3f, 0.3f, 0.3f), c + f3(0.3f, 0.3f, 0.3f));
        lo = vmin(lo, bx.center - bx.half); hi = vmax(hi, bx.center + bx.half);
        b.csg_lo = lo; b.csg_hi = hi;
    }
    F3 rc = 0.5f * (b.nodes[0].lo + b.nodes[0].hi);
    F3 rh = b.nodes[0].hi - rc;
    /* pushes in main()'s order (macos_main.mm:478-538) */
    for (uint32_t i = 0; i < b.tris.size(); ++i) b.insert(0, rc, rh, 0, Shape{S_TRI, i});
    for (uint32_t i = 0; i < scene->cylinders.size(); ++i) b.insert(0, rc, rh, 0, Shape{S_CYL, i});
    for (uint32_t i = 0; i < scene->boxes.size(); ++i) b.insert(0, rc, rh, 0, Shape{S_BOX, i});
    for (uint32_t i = 0; i < scene->spheres.size(); ++i) b.insert(0, rc, rh, 0, Shape{S_SPHERE, i});
    if (scene->reference_csg) b.insert(0, rc, rh, 0, Shape{S_CSG, 0});

    /* ---- flatten for the device --------------------------------------------------------- */
    /* source shape -> device slot (ort_tree.cpp reordered the primitives) */
    auto slot_of = [&](Shape s) -> uint32_t {
        switch (s.type) {
        case S_TRI: return (PRIM_TRI << 28) | tree.tri_slot[s.index];
        case S_SPHERE: return (PRIM_SPHERE << 28) | tree.sphere_slot[s.index];
        case S_BOX: return (PRIM_BOX << 28) | tree.box_slot[s.index];
        default: return (PRIM_CYL << 28) | tree.cyl_slot[s.index];
        }
    };
    out->tri_chain.assign(tree.tris.size(), 0);
    out->sphere_chain.assign(tree.spheres.size(), 0);
    out->box_chain.assign(tree.boxes.size(), 0);
    out->cyl_chain.assign(tree.cyls.size(), 0);
    out->nodes.resize(b.nodes.size());
    for (size_t ni = 0; ni < b.nodes.size(); ++ni) {
        const Node &n = b.nodes[ni];
        DevRefNode &d = out->nodes[ni];
        d.lo[0] = n.lo.x; d.lo[1] = n.lo.y; d.lo[2] = n.lo.z;
        d.hi[0] = n.hi.x; d.hi[1] = n.hi.y; d.hi[2] = n.hi.z;
        d.first_child = n.first_child;
        d.rec_first = (uint32_t)out->recs.size();
        uint32_t live = 0;
        for (Shape s : n.shapes) {
            if (s.type == S_CSG) continue; /* its hit test is compiled out (ray.cpp:718-767) */
            out->recs.push_back(slot_of(s));
            ++live;
        }
        d.rec_count = live;
        /* "has records" in the reference counts the CSG record too (push_buffer.used != 0) */
        d.flags = (n.is_leaf ? 1u : 0u) | (!n.shapes.empty() ? 2u : 0u);
        if (n.shapes.size() > out->max_leaf_records) out->max_leaf_records = (uint32_t)n.shapes.size();
        if (!n.shapes.empty()) out->nonempty_leaves++;

        if (live) {
            /* chain of boxes from this node up to (not including) the root */
            uint32_t first = (uint32_t)(out->chain_boxes.size() / 2);
            uint32_t len = 0;
            for (int32_t a = (int32_t)ni; a > 0; a = b.nodes[(size_t)a].parent) {
                const Node &an = b.nodes[(size_t)a];
                if (len) { /* a node with one occupied child has that child's box: same test, keep one */
                    const F4 &pl = out->chain_boxes[out->chain_boxes.size() - 2], &ph = out->chain_boxes[out->chain_boxes.size() - 1];
                    if (pl.x == an.lo.x && pl.y == an.lo.y && pl.z == an.lo.z && ph.x == an.hi.x && ph.y == an.hi.y && ph.z == an.hi.z) continue;
                }
                out->chain_boxes.push_back(F4{an.lo.x, an.lo.y, an.lo.z, 0});
                out->chain_boxes.push_back(F4{an.hi.x, an.hi.y, an.hi.z, 0});
                ++len;
            }
            if (len > 15 || first > 0x07ffffffu) { *err = "reference octree chain does not fit its encoding"; return ORT_ERR_UNSUPPORTED; }
            /* bit 27: every box lies within the next one up and has lo <= hi (always so for this
               octree -- a shape grows each node it passes -- but the kernel's two-test shortcut
               depends on it, so it is verified rather than assumed) */
            bool nested = true;
            for (uint32_t k = 0; k < len; ++k) {
                const F4 &lo = out->chain_boxes[2u * (first + k)], &hi = out->chain_boxes[2u * (first + k) + 1u];
                if (!(lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z)) nested = false;
                if (k + 1 < len) {
                    const F4 &ulo = out->chain_boxes[2u * (first + k + 1u)], &uhi = out->chain_boxes[2u * (first + k + 1u) + 1u];
                    if (!(lo.x >= ulo.x && lo.y >= ulo.y && lo.z >= ulo.z && hi.x <= uhi.x && hi.y <= uhi.y && hi.z <= uhi.z)) nested = false;
                }
            }
            if (!nested) out->unnested_chains++;
            uint32_t word = (len << 28) | (nested ? 0x08000000u : 0u) | first;
            for (size_t r = d.rec_first; r < out->recs.size(); ++r) {
                uint32_t kind = out->recs[r] >> 28, slot = out->recs[r] & 0x00ffffffu;
                if (kind == PRIM_TRI) out->tri_chain[slot] = word;
                else if (kind == PRIM_SPHERE) out->sphere_chain[slot] = word;
                else if (kind == PRIM_BOX) out->box_chain[slot] = word;
                else out->cyl_chain[slot] = word;
            }
        }
    }
    /* static breadth-first numbering of every record = the order in which the reference tests
       shapes along any ray (unvisited nodes only drop out of the sequence) */
    out->tri_order.assign(tree.tris.size(), 0);
    out->sphere_order.assign(tree.spheres.size(), 0);
    out->box_order.assign(tree.boxes.size(), 0);
    out->cyl_order.assign(tree.cyls.size(), 0);
    {
        std::vector<uint32_t> queue;
        queue.push_back(0);
        uint32_t next = 0;
        for (size_t head = 0; head < queue.size(); ++head) {
            const DevRefNode &d = out->nodes[queue[head]];
            for (uint32_t r = 0; r < d.rec_count; ++r) {
                uint32_t rec = out->recs[d.rec_first + r], kind = rec >> 28, slot = rec & 0x00ffffffu;
                if (kind == PRIM_TRI) out->tri_order[slot] = next;
                else if (kind == PRIM_SPHERE) out->sphere_order[slot] = next;
                else if (kind == PRIM_BOX) out->box_order[slot] = next;
                else out->cyl_order[slot] = next;
                ++next;
            }
            if (d.first_child >= 0)
                for (uint32_t k = 0; k < 8; ++k) queue.push_back((uint32_t)d.first_child + k);
        }
    }
    if (getenv("ORT_DEBUG_CHAINS")) { /* developer knob: the visibility chains of the analytic shapes */
        auto dump = [&](const char *kind, const std::vector<uint32_t> &words, const std::vector<F4> *shape_boxes) {
            for (size_t i = 0; i < words.size(); ++i) {
                uint32_t len = words[i] >> 28, first = words[i] & 0x07ffffffu;
                fprintf(stderr, "chain %s %zu len %u nested %u:", kind, i, len, (words[i] >> 27) & 1u);
                for (uint32_t k = 0; k < len; ++k) {
                    const F4 &lo = out->chain_boxes[2u * (first + k)], &hi = out->chain_boxes[2u * (first + k) + 1u];
                    fprintf(stderr, " [%g %g %g | %g %g %g]", lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                }
                if (shape_boxes) {
                    const F4 &lo = (*shape_boxes)[2 * i], &hi = (*shape_boxes)[2 * i + 1];
                    fprintf(stderr, "  shape [%g %g %g | %g %g %g]", lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                }
                fprintf(stderr, "\n");
            }
        };
        dump("box", out->box_chain, nullptr);
        dump("sphere", out->sphere_chain, nullptr);
        dump("cyl", out->cyl_chain, nullptr);
    }
    out->built = true;
    return ORT_OK;
}

} // namespace ort
