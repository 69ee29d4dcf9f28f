/*
 * ort_scene.h -- host-side scene object behind the opaque ort_scene handle.
 */
#ifndef ORT_SCENE_H
#define ORT_SCENE_H

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ort.h"

namespace ort {

struct HostMesh {
    std::vector<float> vertices;   // 3 * n, world space after placement
    std::vector<uint32_t> indices; // 3 * triangles
    uint32_t mat = 0;
    ort_v3 aabb_min = {0, 0, 0}, aabb_max = {0, 0, 0};
};

/* ---- device-resident layout (mirrored on the host before upload) -----------------------
 * All records are 16-byte multiples so one lane fetches a record with dwordx4 loads.
 *
 * Node (64 B): a binary node carrying BOTH children's boxes, so one fetch decides both.
 *   lo0.xyz hi0.xyz lo1.xyz hi1.xyz child0 child1 pad pad
 * child word: bit31 = leaf.  interior: [29:0] node index, bit30 = a sphere lives below.
 *   leaf: [30:28] primitive kind, [27:24] count-1, [23:0] first index within that kind's array.
 */
struct DevNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    uint32_t child0, child1, pad0, pad1;
};
static_assert(sizeof(DevNode) == 64, "node record must be 64 B");

/* kind codes: SPHERE = 4 so that bit 30 of a leaf word means "sphere leaf"; interior child
   words reuse bit 30 as "a sphere lives below".  Such children are exempt from the
   closer-than-best cull: ray_intersect_with_sphere's tangent branch (ray.cpp:174-183) reports
   t = -b/(2a), a point OUTSIDE the sphere's box and nearer than its entry distance. */
enum : uint32_t { PRIM_TRI = 0, PRIM_BOX = 2, PRIM_CYL = 3, PRIM_SPHERE = 4 };
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t SPHERE_BELOW_BIT = 0x40000000u;
constexpr uint32_t NODE_INDEX_MASK = 0x3fffffffu;
constexpr uint32_t EMPTY_CHILD = 0xffffffffu; /* leaf, kind 7: never visited (box is inverted) */
constexpr uint32_t MAX_LEAF_PRIMS = 16;
/* depth of the fast tree = what a traversal stack must hold at most.  ONE constant: ort_tree.cpp builds within it
   (and refuses a tree that exceeds it), ort_lane.h static_asserts that its smallest stack (the re-traversal of
   resolve_hit: LDS entries minus the four it borrows, plus the scratch tail) holds it */
constexpr uint32_t kTreeDepthBudget = 60;
#ifndef ORT_TREELET_NODES
#define ORT_TREELET_NODES 32
#endif
constexpr uint32_t kTreeletNodes = ORT_TREELET_NODES; /* the breadth-first top of the fast tree has indices [0, 32): kept in LDS by the kernel */

/* Wide node (128 B): the same tree with every other level folded away -- up to FOUR children per node, their boxes
 * stored by coordinate (lo.x of the four children in one 16-byte word, ...) and the four child words; child words as in
 * DevNode, unused slots = EMPTY_CHILD with an inverted box.  A ray then makes half the dependent node fetches on its
 * way down, which is what bounds deep trees (the 1M-triangle scene: ~10 binary visits per ray; profiles/r02_tuning.md).
 * Built from the binary tree by ort_tree.cpp (collapse_to_wide); used by the kernel variant for trees that leave the L2. */
struct DevNode4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(DevNode4) == 128, "wide node record must be 128 B");

inline uint32_t make_leaf(uint32_t kind, uint32_t first, uint32_t count) {
    return LEAF_BIT | (kind << 28) | ((count - 1u) << 24) | (first & 0x00ffffffu);
}

/* Triangle slab (48 B): v0, e1 = v1 - v0, e2 = v2 - v0, n = cross(e1, e2).  e1/e2/n are the
   reference's own f32 expressions (ray.cpp:87-88,110) evaluated once on the host: same
   bits as evaluating them per test. */
struct DevTri { float v0[3], e1[3], e2[3], n[3]; };
static_assert(sizeof(DevTri) == 48, "triangle slab must be 48 B");

struct DevSphere { float c[3]; float r; };                     /* 16 B */
struct DevBox { float lo[3]; float pad0; float hi[3]; float pad1; }; /* 32 B */
/* cylinder: base, radius, the row-major rotation of rotation_matrix_along_z(axis)
   (ray.cpp:8-33) and |axis| (ray.cpp:302), all precomputed with the reference's f32
   expressions; 64 B */
struct DevCyl { float base[3]; float r; float rot[9]; float len; float pad[2]; };
static_assert(sizeof(DevCyl) == 64, "cylinder record must be 64 B");

/* material as the path reads it (ray.h:30-40 minus specular.w) plus the per-material values that
   sample_brdf / pdf_brdf / eval_scattering recompute on every call (ray.cpp:939,1010-1018,
   1105-1113): lobe weights |K|/(|Kd|+|Ks|+|Kt|) and Kd/pi, evaluated once on the host with the
   reference's f32 expressions (same bits); 80 B */
struct DevMaterial {
    float diffuse[3]; float ior;
    float specular[3]; uint32_t is_light;
    float transmission[3]; float pd_c;
    float emit[3]; float ps_c;
    float ed[3]; float pt_c;
};
static_assert(sizeof(DevMaterial) == 80, "material record must be 80 B");
DevMaterial make_dev_material(const ort_material &m); /* ort_tree.cpp */

struct Tree {
    std::vector<DevNode> nodes;
    std::vector<DevNode4> nodes4; /* the 4-wide form of the same tree (empty if it was not built) */
    uint32_t max_depth4 = 0;      /* levels of the wide tree: a traversal stacks at most 3 entries per level */
    std::vector<DevTri> tris;
    std::vector<uint32_t> tri_mat;
    std::vector<DevSphere> spheres;
    std::vector<uint32_t> sphere_mat;
    std::vector<DevBox> boxes;
    std::vector<uint32_t> box_mat;
    std::vector<DevCyl> cyls;
    std::vector<uint32_t> cyl_mat;
    /* source index -> slot in the reordered arrays above (triangles: mesh-major triangle id) */
    std::vector<uint32_t> tri_slot, sphere_slot, box_slot, cyl_slot;
    uint32_t leaf_count = 0, max_leaf_prims = 0, max_depth = 0;
    float sah_cost = 0;
    /* the scene's few analytic shapes are not in the tree: every ray tests all of them up front, the
       whole wave in step (ort_lane.h: prologue_tests); the tree then holds the triangles only */
    bool analytic_prologue = false;
    uint32_t pro_boxes = 0, pro_spheres = 0, pro_cyls = 0; /* shapes [0, n) of each kind form the prologue */
    bool built = false;
};

/* ---- the reference-compatible loose octree (ort_reftree.cpp) ---------------------------
 * Node (48 B): lo.xyz first_child | hi.xyz rec_first | rec_count flags pad pad
 * flags: bit0 = is_leaf, bit1 = the reference's push buffer is non-empty.
 * recs: kind << 28 | slot (same slots as the fast tree's primitive arrays).
 * chain_boxes: per record-holding node, the boxes of that node and its ancestors up to,
 * not including, the root (2 x F4 each); *_chain[slot] = len << 28 | first pair index. */
struct F4 { float x, y, z, w; };
struct DevRefNode {
    float lo[3]; int32_t first_child;
    float hi[3]; uint32_t rec_first;
    uint32_t rec_count, flags, pad0, pad1;
};
static_assert(sizeof(DevRefNode) == 48, "reference octree node must be 48 B");

struct RefTree {
    std::vector<DevRefNode> nodes;
    std::vector<uint32_t> recs;
    std::vector<F4> chain_boxes;
    std::vector<uint32_t> tri_chain, sphere_chain, box_chain, cyl_chain;
    /* position of each primitive in the reference's test order: raycast_bvh walks breadth-first
       (children in slot order) and a node's records in push order, and among bit-equal hit
       distances the first one tested wins (strict <, ray.cpp:653,670,686,708) */
    std::vector<uint32_t> tri_order, sphere_order, box_order, cyl_order;
    uint32_t nonempty_leaves = 0, max_leaf_records = 0;
    uint32_t unnested_chains = 0; /* chains whose boxes are not nested (expected 0; those take the full walk) */
    bool built = false;
};

struct DeviceScene; /* ort_kernels.hip */

struct Scene {
    std::vector<ort_material> materials;
    std::vector<ort_sphere> spheres;
    std::vector<ort_box> boxes;
    std::vector<ort_cylinder> cylinders;
    std::vector<ort_light> lights;
    std::vector<HostMesh> meshes;
    ort_v3 ambient = {0, 0, 0};
    ort_v3 camera_p = {0, 0, 0};
    float camera_quat[4] = {0, 0, 0, 1}; /* xyzw */
    float camera_height_ratio = 0;
    int32_t screen_width = 0, screen_height = 0;

    /* main() inserts one inert CSG shape into its octree (macos_main.mm:322-332,532-538); it can
       never be hit but it shapes the node boxes, so scenes loaded from .scn carry it */
    bool reference_csg = false;

    Tree tree;
    RefTree ref;
    DeviceScene *dev = nullptr;
};

/* ort_parse.cpp */
int parse_scn_text(const char *text, size_t size, const char *base_dir, Scene *scene, std::string *err);
int read_file(const char *path, std::vector<char> *out);
void camera_basis(const Scene &s, int32_t width, int32_t height, ort_camera *out);

/* ort_tree.cpp */
int build_tree(Scene *scene, std::string *err);
/* ort_reftree.cpp */
int build_ref_tree(Scene *scene, std::string *err);

/* ort_hdr.cpp */
uint32_t rgbe_pack(float r, float g, float b);

/* ort_kernels.hip */
int device_count(int *n, std::string *err);
int device_upload(Scene *scene, int device, std::string *err);
void device_release(Scene *scene);
int device_render(Scene *scene, const ort_render_params *p, const ort_tile_job *jobs, uint32_t job_count,
                  void *d_out, float *h_out, void *stream, uint32_t *final_states, ort_stats *stats, std::string *err);
uint64_t render_workspace_bytes(const ort_render_params *p);
int device_unit_eval(int device, const void *records, uint32_t n, float *out, std::string *err);
uint64_t shard_block_count(const ort_render_params *p);
/* ort_comm.cpp */
struct Comm;
uint64_t comm_shard_blocks(int32_t w, int32_t h, uint32_t index, uint32_t count);
void pack_blocks_host(const float *full, int32_t w, int32_t h, uint32_t index, uint32_t count, float *packed);
void unpack_blocks_host(const float *packed, int32_t w, int32_t h, uint32_t index, uint32_t count, float *full);
int unpack_blocks_device(const void *d_packed, int32_t w, int32_t h, uint32_t index, uint32_t count, void *d_full, void *stream, std::string *err);
int comm_unique_id(void *id, std::string *err);
int comm_create(const void *id, int rank, int world, int device, Comm **out, std::string *err);
int comm_create_local(int world, const int *devices, Comm **out, std::string *err);
void comm_destroy(Comm *c);
int gather_framebuffer(Comm *c, const void *d_packed, void *d_full, int32_t w, int32_t h, void *stream, std::string *err);
int gather_framebuffer_local(Comm **cs, int world, const void *const *d_packed, void *d_full_root, int32_t w, int32_t h,
                             void *const *streams, std::string *err);
/* ort_tree.cpp: rotation_matrix_along_z(axis) rows + |axis| as the kernel receives them */
void cylinder_frame_for(const ort_cylinder &c, float rot[9], float *len);

} // namespace ort

struct ort_scene : public ort::Scene {};
struct ort_comm; /* = ort::Comm behind the C ABI */

#endif
