/*
 * include/ort.h -- C ABI of the MI355X-native render path (libort.so).
 *
 * Drop-in boundary for ONE path of gyuhyun-lee/offline_raytracer: the per-pixel path
 * trace `tiled_raytrace_bvh` (reference code/ray.cpp:1178-1466) and the host code on
 * either side of it that the reference keeps in main() (code/macos_main.mm).  The
 * reference has no FFI of its own; each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only.  The compute entry points
 * need a gfx950 device and fail with ORT_ERR_NO_DEVICE / ORT_ERR_HIP otherwise: there
 * is no CPU fallback for the render call.
 *
 * Framebuffer layout everywhere: packed f32 RGB, 12 B per pixel, row-major, row 0 =
 * BOTTOM of the image (ray.cpp:1215-1216; the HDR writer flips, macos_main.mm:686-704).
 */
#ifndef ORT_H
#define ORT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORT_ABI_VERSION 3

enum {
    ORT_OK = 0,
    ORT_ERR_INVALID = 1,     /* bad argument (null, empty rect, spp % chunk, ...) */
    ORT_ERR_IO = 2,          /* file cannot be opened / written */
    ORT_ERR_PARSE = 3,       /* where the reference would trip an assert (platform.h:16-20) */
    ORT_ERR_NO_DEVICE = 4,   /* no HIP device / scene not uploaded */
    ORT_ERR_HIP = 5,         /* a HIP call failed; see ort_last_error() */
    ORT_ERR_UNSUPPORTED = 6, /* e.g. OBJ v/vt faces (parser.cpp:921-923) */
    ORT_ERR_STATE = 7,       /* call order (render before commit/upload) */
    ORT_ERR_NO_MEMORY = 8,   /* a host allocation failed */
    ORT_ERR_INTERNAL = 9     /* an exception the C++ side did not expect; see ort_last_error() */
};

/* thread-local description of the last failure on this thread */
const char *ort_last_error(void);
int ort_abi_version(void);

/* ---- plain data mirrors of the reference structs ---------------------------------- */
typedef struct { float x, y, z; } ort_v3;

/* ray.h:30-40 */
typedef struct {
    ort_v3 diffuse;
    float specular[4]; /* .w ("alpha") is parsed but unused by the path */
    ort_v3 transmission;
    float ior;
    ort_v3 emit;
    int32_t is_light;
} ort_material;

typedef struct { ort_v3 center; float r; uint32_t mat; } ort_sphere;        /* ray.h:4-10 */
typedef struct { ort_v3 min, max; uint32_t mat; } ort_box;                  /* ray.h:12-18 */
typedef struct { ort_v3 base, axis; float r; uint32_t mat; } ort_cylinder;  /* ray.h:20-27 */
typedef struct { uint32_t type; uint32_t index; } ort_light;  /* light push buffer entry: 1 = sphere, 2 = cylinder (ray.h:97-106) */
typedef struct { ort_v3 p, x_axis, y_axis, z_axis; } ort_camera;            /* ray.h:42-49 */

/* ray.h:51-65: placed (world-space) vertices */
typedef struct {
    const float *vertices;
    uint32_t vertex_count;
    const uint32_t *indices;
    uint32_t index_count;
    uint32_t mat;
    ort_v3 aabb_min, aabb_max;
} ort_mesh;

typedef struct {
    const ort_material *materials; uint32_t material_count; /* index 0 = reserved "no hit" */
    const ort_sphere *spheres;     uint32_t sphere_count;
    const ort_box *boxes;          uint32_t box_count;
    const ort_cylinder *cylinders; uint32_t cylinder_count;
    const ort_mesh *meshes;        uint32_t mesh_count;
    const ort_light *lights;       uint32_t light_count;
    /* camera as parsed (parser.cpp:1208-1226) */
    ort_v3 camera_p;
    float camera_quat_xyzw[4];
    float camera_height_ratio;
    int32_t screen_width, screen_height;
    ort_v3 ambient;
    /* main() pushes one inert CSG shape into its octree (macos_main.mm:322-332,532-538).  It is
       never hit, but it shapes the octree node boxes, and those decide which shapes a ray that
       starts exactly on a node face can see (ray.cpp:788-803).  Non-zero: reproduce it, as
       ort_scene_load_scn always does. */
    int32_t with_reference_csg;
} ort_scene_desc;

typedef struct {
    uint32_t material_count, sphere_count, box_count, cylinder_count, mesh_count, light_count;
    uint32_t triangle_count;
    int32_t screen_width, screen_height; /* the .scn "screen" line */
    ort_v3 ambient;
    ort_v3 camera_p;
    float camera_quat_xyzw[4];
    float camera_height_ratio;
} ort_scene_info;

typedef struct {
    uint32_t node_count, leaf_count, max_leaf_prims, max_depth;
    uint64_t node_bytes, prim_bytes; /* resident in HBM after upload */
    float sah_cost;
    /* the reference-compatible loose octree kept beside it (visibility chains + fallback) */
    uint32_t ref_node_count, ref_nonempty_leaves, ref_max_leaf_records;
    uint64_t ref_bytes;
    /* analytic shapes kept out of the tree and tested outright by every ray (0: all shapes are in the tree) */
    uint32_t prologue_prims;
    /* the 4-wide form of the same tree (128-byte nodes, up to four children each), which renders of trees that do not
       fit the L2 traverse instead: half the dependent node fetches per ray */
    uint32_t wide_node_count, wide_max_depth;
} ort_tree_info;

/* work counters of one render call (device counters).  All zero unless
   ORT_RENDER_COUNTERS is passed. */
typedef struct {
    uint64_t paths, rays, node_tests, tri_tests, analytic_tests;
    uint64_t fallback_rays; /* rays re-cast on the reference-compatible octree (Exactness) */
    double kernel_ms; /* HIP-event time of the path-trace kernel(s), always filled */
} ort_stats;

typedef struct ort_scene ort_scene;

/* ---- scene ingestion (host) ---------------------------------------------------------
 * ort_scene_load_scn replaces parse_scene (parser.cpp:1184-1446) + the mesh loading and
 * placement loop of main() (macos_main.mm:342-414): .scn grammar, ASCII PLY
 * (parser.cpp:384-570) and OBJ (parser.cpp:687-982) with the reference's number lexer
 * (parser.cpp:158-250), fan triangulation and placement arithmetic, bit for bit.
 * base_dir is prepended to mesh file names exactly as given (parser.cpp:1436-1438). */
int ort_scene_load_scn(const char *scn_path, const char *base_dir, ort_scene **out);
/* the same from memory-resident text (scn_text need not be NUL-terminated) */
int ort_scene_parse_scn(const char *scn_text, size_t scn_size, const char *base_dir, ort_scene **out);
/* scene from flattened arrays (copied) */
int ort_scene_create(const ort_scene_desc *desc, ort_scene **out);
void ort_scene_destroy(ort_scene *scene);

int ort_scene_get_info(const ort_scene *scene, ort_scene_info *out);
/* copy-out accessors; cap = capacity of out in elements; returns ORT_ERR_INVALID if too small */
int ort_scene_get_materials(const ort_scene *scene, ort_material *out, uint32_t cap);
int ort_scene_get_spheres(const ort_scene *scene, ort_sphere *out, uint32_t cap);
int ort_scene_get_boxes(const ort_scene *scene, ort_box *out, uint32_t cap);
int ort_scene_get_cylinders(const ort_scene *scene, ort_cylinder *out, uint32_t cap);
int ort_scene_get_lights(const ort_scene *scene, ort_light *out, uint32_t cap);
/* borrowed pointers, valid until ort_scene_destroy */
int ort_scene_get_mesh(const ort_scene *scene, uint32_t mesh_index, ort_mesh *out);
/* camera basis for a W x H image: macos_main.mm:550-556 */
int ort_scene_get_camera(const ort_scene *scene, int32_t width, int32_t height, ort_camera *out);

/* ---- acceleration structure (host) --------------------------------------------------
 * Replaces the octree construction of main() (macos_main.mm:418-545 driving
 * ray.cpp:1799-2045).  Closest-hit results do not depend on the tree (SURVEY 8a note),
 * so this builds the GPU layout directly: a flat SoA tree with 16-byte-aligned node
 * records and triangle slabs. */
int ort_scene_commit(ort_scene *scene);
int ort_scene_get_tree_info(const ort_scene *scene, ort_tree_info *out);

/* ---- device ---------------------------------------------------------------------- */
int ort_device_count(int *count);
/* copies the committed scene into the HBM of HIP device <device> (hipSetDevice) */
int ort_scene_upload(ort_scene *scene, int device);

/* ---- the render call ---------------------------------------------------------------
 * One job == one call of the reference function
 *   u64 tiled_raytrace_bvh(World*, Camera*, BVHOctreeNode*, v3 *out, i32 W, i32 H,
 *                          i32 x0, i32 y0, i32 x1, i32 y1, RandomSeries*, u32 spp, f32 rr)
 * (ray.cpp:1178-1183): the pixels of [x0,x1) x [y0,y1) are rendered serially, row-major,
 * with ONE xorshift stream threaded through every pixel and sample of the rect. */
typedef struct {
    int32_t x0, y0, x1, y1;
    uint32_t rng_state; /* RandomSeries.next_random on entry (random.h:17-29) */
    uint32_t spp;
} ort_tile_job;

/* exact analogue of a single reference call; *rng_state is updated as the reference
   updates its RandomSeries.  out_rgb is a HOST buffer of width*height*3 floats; only the
   rect is written.  shape_tests (may be NULL) receives this tree's intersection-test
   count (the reference's return value is tree-dependent diagnostics). */
int ort_tiled_raytrace(ort_scene *scene, float *out_rgb, int32_t width, int32_t height, int32_t x0, int32_t y0,
                       int32_t x1, int32_t y1, uint32_t *rng_state, uint32_t spp, float rr, uint64_t *shape_tests);

/* many reference calls in one launch (one GPU lane per job).  rects must be disjoint.
   final_states (may be NULL) receives each job's final RNG state. */
int ort_tiled_raytrace_batch(ort_scene *scene, float *out_rgb, int32_t width, int32_t height,
                             const ort_tile_job *jobs, uint32_t job_count, float rr, uint32_t *final_states,
                             ort_stats *stats);

/* Seeding policies = the caller side of the reference call (main()'s tile loop,
 * macos_main.mm:602-662).  job_seed(m, j) = fmix32(m ^ (j * 2654435761u)), 0 -> 1.
 *  TILE32 : main()'s schedule: 32x32 tiles of ceil(W/32) x ceil(H/32) px, tile series =
 *           random_u32(&master) in row-major tile order, master state = seed.
 *  WHOLE  : one call over the rect, series = random_u32(&master).
 *  PIXEL  : one call per pixel (1x1 rect, all spp), series = job_seed(seed, y*W + x).
 *  CHUNK  : spp/chunk calls per pixel of <chunk> samples; call k of pixel i uses
 *           job_seed(seed, k*W*H + i); pixel = (sum_k call_k, in k order) / (spp/chunk).
 * Results are independent of how pixels are spread over lanes, workgroups or GPUs. */
enum { ORT_POLICY_TILE32 = 0, ORT_POLICY_WHOLE = 1, ORT_POLICY_PIXEL = 2, ORT_POLICY_CHUNK = 3 };

/* flags.  ORT_RENDER_PACKED (PIXEL / CHUNK policies): the output buffer holds only this shard's 8x8 blocks,
   [local block k][pixel in block, row-major 8x8][rgb] with k-th block = block id shard_index + k * shard_count of
   the row-major block grid (ceil(W/8) wide, row 0 = bottom); ort_shard_block_count blocks, 768 B each.  That is
   the layout ort_gather_framebuffer moves between GPUs. */
enum { ORT_RENDER_COUNTERS = 1, ORT_RENDER_PACKED = 2 };

typedef struct {
    int32_t width, height;
    int32_t x0, y0, x1, y1; /* pixels to render (clipped to the image) */
    int32_t policy;
    uint32_t seed;
    uint32_t spp;
    uint32_t chunk; /* ORT_POLICY_CHUNK only */
    float rr;       /* russian roulette continue probability; main() uses 0.8 */
    uint32_t flags;
    /* multi-GPU sharding of PIXEL / CHUNK renders: the image is cut into 8x8-pixel
       blocks numbered row-major; this call renders blocks with id % shard_count ==
       shard_index.  shard_count <= 1: everything. */
    uint32_t shard_index, shard_count;
} ort_render_params;

/* host framebuffer in, host framebuffer out (device staging is internal); synchronous */
int ort_render_image(ort_scene *scene, const ort_render_params *params, float *out_rgb, ort_stats *stats);

/* device framebuffer: d_out_rgb is a DEVICE pointer (width*height*3 floats) on the scene's
   device, e.g. a torch tensor's data_ptr.  Work is enqueued on hip_stream (a hipStream_t
   passed as void*, NULL = the default stream) and the call returns without waiting
   unless stats != NULL.  Pixels of other shards are left untouched. */
int ort_render_image_device(ort_scene *scene, const ort_render_params *params, void *d_out_rgb, void *hip_stream,
                            ort_stats *stats);

/* bytes of device workspace ort_render_image_device keeps for these params (CHUNK partial sums, held in the packed
   block layout: a shard keeps 1/shard_count of a frame per chunk) */
int ort_render_workspace_bytes(const ort_render_params *params, uint64_t *bytes);

/* ---- multi-GPU: block sharding and the one collective -------------------------------------
 * Replaces main()'s shared-memory tile pool (macos_main.mm:565-671: eight pthreads, one queue, one framebuffer)
 * across the GPUs of a node: scene replicated, 8x8 blocks dealt round-robin, every rank renders its blocks into a
 * packed buffer (ORT_RENDER_PACKED) and ONE gather brings them to rank 0 -- grouped ncclSend / ncclRecv over RCCL
 * (xGMI), then an un-permute kernel on rank 0.  Seeds are per pixel, so the assembled image is bit-identical to a
 * one-GPU render.  librccl.so is loaded on first use. */
int ort_shard_block_count(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, uint64_t *blocks);
/* host-side (CPU) packing, for callers that move the blocks themselves */
int ort_pack_blocks_host(const float *full_rgb, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *packed);
int ort_unpack_blocks_host(const float *packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *full_rgb);
/* device-side un-permute of one shard's packed blocks into a full frame (both DEVICE pointers) */
int ort_unpack_blocks_device(const void *d_packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count,
                             void *d_full_rgb, void *hip_stream);

typedef struct ort_comm ort_comm;
#define ORT_COMM_ID_BYTES 128
/* one process per GPU: rank 0 draws an id (ncclGetUniqueId), hands it to the other ranks by any means, then every
   rank creates its communicator on its device.  world == 1 needs no id and never touches RCCL. */
int ort_comm_unique_id(void *id /* ORT_COMM_ID_BYTES */);
int ort_comm_create(const void *id, int rank, int world, int device, ort_comm **out);
/* one process driving all GPUs (ncclCommInitAll): fills out[0 .. world) */
int ort_comm_create_local(int world, const int *devices, ort_comm **out);
void ort_comm_destroy(ort_comm *comm);
/* the collective: every rank passes its packed blocks (device pointer, same width/height/world as rendered with);
   rank 0 also passes the full frame to assemble (device pointer, width*height*3 floats), others NULL.  Enqueued on
   hip_stream of the communicator's device; returns without waiting. */
int ort_gather_framebuffer(ort_comm *comm, const void *d_packed, void *d_full_rgb, int32_t width, int32_t height, void *hip_stream);
/* the same for ort_comm_create_local's communicators, all ranks in one call (streams may be NULL) */
int ort_gather_framebuffer_local(ort_comm **comms, int world, const void *const *d_packed, void *d_full_rgb_rank0, int32_t width,
                                 int32_t height, void *const *hip_streams);

/* ---- diagnostics: per-function evaluation ON THE DEVICE (parity tests) -----------------
 * records: count x {u32 op; f32 in[24]}; out: count x f32[8].  Ops (reference file:line):
 *  1 triangle  ray.cpp:63-115   in v0 v1 v2 o d              out t n.xyz
 *  2 sphere    ray.cpp:132-190  in c r o d                   out t n.xyz
 *  3 aab       ray.cpp:206-283  in min max o d               out t n.xyz
 *  4 cylinder  ray.cpp:286-352  in base axis r o d           out t n.xyz
 *  5 sample_brdf ray.cpp:1100   in seed(bits) N wo rough Kd Ks Kt ior   out wi.xyz is_transmission rng(bits)
 *  6 pdf_brdf  ray.cpp:1007     in N wi wo rough Kd Ks Kt ior           out p
 *  7 eval_scattering ray.cpp:936 in N wi wo Kd Ks Kt ior rough dist     out f.xyz
 *  8 sample_lobe ray.cpp:1065   in N c phi                   out v.xyz
 *  9 libm      (deterministic)  in x y                       out sinf(x) cosf(x) atan2f(y,x) powf(x,y) logf(x)
 * 10 normalize math.h:298-310   in v                         out v.xyz
 * 11 fresnel/ggx/geometry ray.cpp:825-897 in Ks l_dot_h N H rough w     out F.xyz D G
 * 12 rng       random.h:5-53    in seed(bits) job(bits)      out step(bits) rng_01 rng_between(0,2pi) state(bits) job_seed(bits)
 * 13 IEEE ops                   in a b c                     out a/b sqrt(a) a*b a+b a-b (f32)bits(a) a*b+c */
int ort_unit_eval_device(int device, const void *records, uint32_t count, float *out);

/* ---- output ------------------------------------------------------------------------
 * v3_to_rgbe (macos_main.mm:242-261) and the .hdr writer (macos_main.mm:263-287,683-707) */
uint32_t ort_rgbe(float r, float g, float b);
int ort_write_hdr(const char *path, const float *rgb, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif /* ORT_H */
