/*
 * oracle/ort_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (offline_raytracer_amd/) never links,
 * imports or calls it.
 *
 * Parity status: PINNED.  Every function is checked bit for bit against the
 * reference's own code compiled in the dev container (oracle/_ref/ref_det, see
 * oracle/ref_driver.cpp) through the committed fixtures under tests/golden/.
 * libm calls go through oracle/det_math.h on both sides (see that header for why).
 */
#ifndef ORT_ORACLE_H
#define ORT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } o_v3;

/* ray.h:30-40 (specular.w is parsed but never read by the path) */
typedef struct {
    o_v3 diffuse;
    float specular[4];
    o_v3 transmission;
    float ior;
    o_v3 emit;
    int32_t is_light;
} o_material;

typedef struct { o_v3 center; float r; uint32_t mat; } o_sphere;       /* ray.h:4-10  */
typedef struct { o_v3 min, max; uint32_t mat; } o_box;                 /* ray.h:12-18 */
typedef struct { o_v3 base, axis; float r; uint32_t mat; } o_cylinder; /* ray.h:20-27 */

/* ray.h:51-65; vertices already placed in world space */
typedef struct {
    const float *vertices; /* 3 * vertex_count */
    uint32_t vertex_count;
    const uint32_t *indices;
    uint32_t index_count;
    uint32_t mat;
    o_v3 aabb_min, aabb_max;
} o_mesh;

/* one entry of the light push buffer (parser.cpp:1144-1182): reference ShapeType
   value (1 = sphere, 2 = cylinder) + index into that array */
typedef struct { uint32_t type; uint32_t index; } o_light;

typedef struct { o_v3 p, x_axis, y_axis, z_axis; } o_camera; /* ray.h:42-49 */

typedef struct {
    const o_material *materials; uint32_t material_count;
    const o_sphere *spheres;     uint32_t sphere_count;
    const o_box *boxes;          uint32_t box_count;
    const o_cylinder *cylinders; uint32_t cylinder_count;
    const o_mesh *meshes;        uint32_t mesh_count;
    const o_light *lights;       uint32_t light_count;
    int32_t with_reference_csg; /* insert main()'s inert CSG shape (macos_main.mm:322-332,532-538) */
    uint32_t octree_depth;      /* macos_main.mm:474 uses 10 */
} o_scene_desc;

typedef struct o_scene o_scene;

typedef struct {
    uint32_t nodes, nonempty_leaves, max_leaf_records;
    uint64_t record_bytes; /* in the reference's own record sizes */
} o_tree_stats;

/* per-render work counters: exact, from the restated reference traversal */
typedef struct {
    uint64_t paths, rays, node_pops, child_tests, tri_tests, analytic_tests, shapes_tested;
    uint32_t final_rng;
    double seconds;
} o_render_stats;

enum { O_POLICY_TILE32 = 0, O_POLICY_WHOLE = 1, O_POLICY_PIXEL = 2, O_POLICY_CHUNK = 3 };

/* scene assembly: copies the arrays, builds the reference's loose octree
   (ray.cpp:1468-2045 driven as in macos_main.mm:418-545) */
o_scene *oracle_scene_create(const o_scene_desc *desc);
void oracle_scene_destroy(o_scene *scene);
void oracle_tree_stats(const o_scene *scene, o_tree_stats *out);

/* ray.cpp:1178-1466.  out is W*H*3 floats, row 0 = bottom.  Returns shapes tested. */
uint64_t oracle_tiled_raytrace(const o_scene *scene, const o_camera *camera, float *out, int32_t width, int32_t height,
                               int32_t x0, int32_t y0, int32_t x1, int32_t y1, uint32_t *rng_state, uint32_t spp,
                               float rr, o_render_stats *stats_accum);

/* the callers' seeding policies (macos_main.mm:602-662 for TILE32; oracle/ref_driver.cpp
   cmd_render documents the others).  threads >= 1 (work split by job; results do not
   depend on the thread count). */
int oracle_render_image(const o_scene *scene, const o_camera *camera, float *out, int32_t width, int32_t height,
                        int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t policy, uint32_t seed, uint32_t spp,
                        uint32_t chunk, float rr, int32_t threads, o_render_stats *stats);

uint32_t oracle_job_seed(uint32_t master, uint32_t job);

/* ray.cpp:1165-1176 + 624-822 */
void oracle_raycast(const o_scene *scene, const float origin[3], const float dir[3], float *t, float normal[3],
                    uint32_t *mat);

/* per-function tables; op codes and layouts as in oracle/ref_driver.cpp cmd_unit */
void oracle_unit(uint32_t op, const float in[24], float out[8]);
void oracle_unit_batch(const uint8_t *records, uint64_t n, float *out);

/* random.h:5-117: n x {state, f32} then derived draws, same layout as ref_driver rng */
void oracle_rng_table(uint32_t seed, uint32_t n, uint8_t *out);

/* macos_main.mm:242-261 and 263-287,683-707 */
uint32_t oracle_rgbe(float r, float g, float b);
int oracle_write_hdr(const char *path, const float *rgb, int32_t width, int32_t height);

/* macos_main.mm:550-556 */
void oracle_camera(const float p[3], const float quat_xyzw[4], float height_ratio, int32_t width, int32_t height,
                   o_camera *out);

#ifdef __cplusplus
}
#endif
#endif
