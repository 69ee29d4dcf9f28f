/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE.  Never shipped, never linked into the product.
 *
 * A Linux command-line driver around the REFERENCE's own hot path.  The reference
 * sources are compiled where they lie (/root/reference/code, via -iquote; see
 * oracle/Makefile) -- nothing of them is copied into this repository.  This file
 * replaces only code/macos_main.mm (the macOS-bound translation unit: mach/Carbon/
 * OSAtomic/dispatch headers, vm_allocate, hard-coded scene path and resolution,
 * time-based seed), and re-expresses in its own words the scene assembly that
 * main() performs before rendering (macos_main.mm:312-562): light arena + parse,
 * the inert hard-coded CSG shape, mesh load + placement, root AABB, octree pushes in
 * the order meshes/cylinders/boxes/spheres/CSG, compaction, camera basis.
 *
 * Two binaries are built from it (oracle/Makefile):
 *   _ref/ref_glibc : reference + glibc libm ("as shipped" on this box)
 *   _ref/ref_det   : reference + oracle/det_math.h interposed for sinf cosf atan2f
 *                    powf logf (link-time: the executable's own definitions win over
 *                    libm.so), built with -fno-builtin.  This is the bit-exact parity
 *                    anchor for the CPU restatement and for the HIP kernel.
 *
 * Commands (all binary I/O is little-endian, formats documented in
 * tests/golden/README.md and mirrored by tests/ref_io.py):
 *   scene-dump <scn> <base_dir> <W> <H> <out.bin>
 *   render <scn> <base_dir> <W> <H> <spp> <seed> <policy> <out.f32> [chunk] [x0 y0 x1 y1]
 *          policy = tile32 | whole | pixel | chunk | sample
 *   unit <in.bin> <out.bin>        per-function input/output tables
 *   rng <seed> <n> <out.bin>
 *   raycast <scn> <base_dir> <rays.bin> <out.bin>
 */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

/* same order as macos_main.mm:14-22 */
#include "types.h"
#include "math.h"
#include "platform.h"
#include "intrinsic.h"
#include "random.h"
#include "ray.cpp"
#include "parser.cpp"

#ifdef ORT_REF_DETMATH
#undef sin
#undef cos
#undef tan
#undef acos
#undef atan2
extern "C" {
#include "det_math.h"
/* Definitions in the executable take precedence over libm.so's at link time. */
float sinf(float x) __THROW { return det_sinf(x); }
float cosf(float x) __THROW { return det_cosf(x); }
float atan2f(float y, float x) __THROW { return det_atan2f(y, x); }
float powf(float x, float y) __THROW { return det_powf(x, y); }
float logf(float x) __THROW { return det_logf(x); }
}
#endif

static PlatformReadFileResult
read_whole_file(const char *path)
{
    PlatformReadFileResult r = {};
    int fd = open(path, O_RDONLY);
    if (fd < 0) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    struct stat st;
    fstat(fd, &st);
    r.size = (u64)st.st_size;
    r.memory = (u8 *)malloc(r.size + 16);
    memset(r.memory, 0, r.size + 16);
    u64 got = 0;
    while (got < r.size) {
        ssize_t n = read(fd, r.memory + got, r.size - got);
        if (n <= 0) break;
        got += (u64)n;
    }
    close(fd);
    return r;
}

/* Everything main() owns between parse and the tile loop. */
struct RefScene
{
    ParseSceneResult parsed;
    World world;
    Camera camera;
    Mesh meshes[100];
    u32 mesh_count;
    CSG csgs[10];
    u32 csg_count;
    BVHOctreeNode *root;
    MemoryArena light_arena, node_arena, shape_arena;
    i32 width, height;
};

static RefScene *g_scene;

static void
assemble_scene(const char *scn_path, const char *base_dir, i32 width, i32 height)
{
    RefScene *s = (RefScene *)calloc(1, sizeof(RefScene));
    g_scene = s;

    /* one zeroed block carved into light / node / shape arenas, consecutively
       (macos_main.mm:302-312,418-419,540-541) */
    size_t light_size = megabytes(2), node_size = megabytes(512), shape_size = megabytes(8);
    u8 *block = (u8 *)calloc(1, light_size + node_size + shape_size + 4096);
    s->light_arena = start_memory_arena(block, light_size);
    s->parsed.light_push_buffer = start_temp_memory(&s->light_arena, s->light_arena.total_size);

    PlatformReadFileResult scn = read_whole_file(scn_path);
    parse_scene(&s->parsed, scn.memory, (u32)scn.size, (char *)base_dir);
    s->parsed.output_width = width;
    s->parsed.output_height = height;
    s->width = width;
    s->height = height;

    /* the hard-coded CSG instance (macos_main.mm:322-332): material 5, a sphere of
       radius 0.35 and a box of half-size 0.3 around (0,0,0.8).  Its hit test is
       compiled out in ray.cpp, so it only shapes the octree. */
    v3 csg_c = v3_(0, 0, 0.8f);
    s->csgs[0].sphere.center = csg_c;
    s->csgs[0].sphere.r = 0.35f;
    s->csgs[0].aab.min = csg_c - v3_(0.3f, 0.3f, 0.3f);
    s->csgs[0].aab.max = csg_c + v3_(0.3f, 0.3f, 0.3f);
    s->csgs[0].mat_index = 5;
    s->csg_count = 1;

    s->world.ambient = s->parsed.ambient;
    s->world.materials = s->parsed.materials;
    s->world.mat_count = s->parsed.mat_count;
    s->world.light_push_buffer = s->parsed.light_push_buffer;
    s->world.light_count = s->parsed.light_count;

    /* mesh load + placement (macos_main.mm:342-414) */
    for (u32 mi = 0; mi < s->parsed.mesh_count; ++mi) {
        MeshInfo *info = s->parsed.mesh_infos + mi;
        Mesh *mesh = s->meshes + s->mesh_count++;
        char ext[16] = {};
        get_extension(ext, info->file_path);
        PlatformReadFileResult f = read_whole_file(info->file_path);
        if (string_compare(ext, (char *)"ply")) {
            ParsePlyHeaderResult h = parse_ply_header(f.memory, (u32)f.size);
            mesh->vertex_count = h.vertex_count;
            mesh->index_count = h.index_count;
            mesh->vertices = (v3 *)malloc(sizeof(v3) * (mesh->vertex_count + 1));
            mesh->indices = (u32 *)malloc(sizeof(u32) * (mesh->index_count + 1));
            mesh->mat_index = info->mat_index;
            parse_ply(&h, f.memory, (u32)f.size, mesh->vertices, mesh->indices);
        } else if (string_compare(ext, (char *)"obj")) {
            PreParseObjResult pre = pre_parse_obj(f.memory, f.size);
            mesh->vertex_count = pre.position_count;
            mesh->index_count = pre.index_count;
            mesh->vertices = (v3 *)malloc(sizeof(v3) * (mesh->vertex_count + 1));
            mesh->indices = (u32 *)malloc(sizeof(u32) * (mesh->index_count + 1));
            mesh->mat_index = info->mat_index;
            parse_obj(&pre, f.memory, (u32)f.size, mesh->vertices, 0, 0, mesh->indices);
        } else {
            fprintf(stderr, "unknown mesh extension '%s'\n", ext);
            exit(2);
        }

        v3 lo = v3_(Flt_Max, Flt_Max, Flt_Max);
        v3 hi = v3_(Flt_Min, Flt_Min, Flt_Min);
        for (u32 vi = 0; vi < mesh->vertex_count; ++vi) {
            v3 *v = mesh->vertices + vi;
            *v *= info->scale;
            *v = quaternion_rotation(info->quaternion,
                                     quaternion_rotation(v3_(0, 1, 0), 0.0174533f * info->axis.degree, *v));
            *v += info->translate;
            lo = gather_min_elements(lo, *v);
            hi = gather_max_elements(hi, *v);
        }
        if (mesh->vertex_count) { /* the reference assigns these inside the vertex loop */
            mesh->aabb_min = lo;
            mesh->aabb_max = hi;
        }
    }

    /* octree root and its AABB (macos_main.mm:418-472) */
    s->node_arena = start_memory_arena(block + light_size, node_size);
    s->root = push_struct(&s->node_arena, BVHOctreeNode);
    zero(s->root);
    s->root->aabb_min = v3_(Flt_Max, Flt_Max, Flt_Max);
    s->root->aabb_max = v3_(Flt_Min, Flt_Min, Flt_Min);
    for (u32 i = 0; i < s->parsed.mesh_count; ++i)
        update_aabb_min_max(&s->root->aabb_min, &s->root->aabb_max, s->meshes + i, Shape_Type_Mesh);
    for (u32 i = 0; i < s->parsed.cylinder_count; ++i)
        update_aabb_min_max(&s->root->aabb_min, &s->root->aabb_max, s->parsed.cylinders + i, Shape_Type_Cylinder);
    for (u32 i = 0; i < s->parsed.box_count; ++i)
        update_aabb_min_max(&s->root->aabb_min, &s->root->aabb_max, s->parsed.boxes + i, Shape_Type_AAB);
    for (u32 i = 0; i < s->parsed.sphere_count; ++i)
        update_aabb_min_max(&s->root->aabb_min, &s->root->aabb_max, s->parsed.spheres + i, Shape_Type_Sphere);
    for (u32 i = 0; i < s->csg_count; ++i) {
        CSG *c = s->csgs + i;
        c->aabb_min = Flt_Max * v3_(1, 1, 1);
        c->aabb_max = Flt_Min * v3_(1, 1, 1);
        update_aabb_min_max(&c->aabb_min, &c->aabb_max, &c->sphere, Shape_Type_Sphere);
        update_aabb_min_max(&c->aabb_min, &c->aabb_max, &c->aab, Shape_Type_AAB);
    }
    v3 root_center = 0.5f * (s->root->aabb_min + s->root->aabb_max);
    v3 root_half = s->root->aabb_max - root_center;
    const u32 depth_limit = 10; /* macos_main.mm:474 */

    /* pushes, in main()'s order: triangles of every mesh, cylinders, boxes, spheres, CSG */
    for (u32 mi = 0; mi < s->mesh_count; ++mi) {
        Mesh *mesh = s->meshes + mi;
        for (u32 k = 0; k + 2 < mesh->index_count; k += 3) {
            Triangle t = {};
            t.mesh = mesh;
            t.i_0 = mesh->indices[k];
            t.i_1 = mesh->indices[k + 1];
            t.i_2 = mesh->indices[k + 2];
            push_shape_inside_node(&s->node_arena, s->root, root_center, root_half, 0, depth_limit, &t, Shape_Type_Triangle);
        }
    }
    for (u32 i = 0; i < s->parsed.cylinder_count; ++i)
        push_shape_inside_node(&s->node_arena, s->root, root_center, root_half, 0, depth_limit, s->parsed.cylinders + i, Shape_Type_Cylinder);
    for (u32 i = 0; i < s->parsed.box_count; ++i)
        push_shape_inside_node(&s->node_arena, s->root, root_center, root_half, 0, depth_limit, s->parsed.boxes + i, Shape_Type_AAB);
    for (u32 i = 0; i < s->parsed.sphere_count; ++i)
        push_shape_inside_node(&s->node_arena, s->root, root_center, root_half, 0, depth_limit, s->parsed.spheres + i, Shape_Type_Sphere);
    for (u32 i = 0; i < s->csg_count; ++i)
        push_shape_inside_node(&s->node_arena, s->root, root_center, root_half, 0, depth_limit, s->csgs + i, Shape_Type_CSG);

    s->shape_arena = start_memory_arena(block + light_size + node_size, shape_size);
    ValidateNodesResult vr = {};
    validate_nodes_and_reallocate_shapes(&s->shape_arena, s->root, &vr);

    /* camera basis (macos_main.mm:550-556) */
    s->camera.p = s->parsed.camera_p;
    f32 rx = s->parsed.camera_height_ratio * ((f32)s->parsed.output_width / s->parsed.output_height);
    s->camera.x_axis = rx * quaternion_rotation(s->parsed.camera_quaternion, v3_(1, 0, 0));
    s->camera.y_axis = s->parsed.camera_height_ratio * quaternion_rotation(s->parsed.camera_quaternion, v3_(0, 1, 0));
    s->camera.z_axis = quaternion_rotation(s->parsed.camera_quaternion, v3_(0, 0, 1));
}

/* ---- small output helpers ------------------------------------------------------- */
static void put_u32(FILE *f, u32 v) { fwrite(&v, 4, 1, f); }
static void put_f32(FILE *f, f32 v) { fwrite(&v, 4, 1, f); }
static void put_v3(FILE *f, v3 v) { fwrite(&v, 12, 1, f); }

static u32
fmix32(u32 h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
/* seed of job j under master seed m; never 0 (0 is a fixed point of xor_shift_32) */
static u32
job_seed(u32 master, u32 j)
{
    u32 h = fmix32(master ^ (j * 2654435761u));
    return h ? h : 1u;
}

static void
count_nodes(BVHOctreeNode *n, u32 *nodes, u32 *leaves, u64 *bytes, u32 *max_leaf)
{
    (*nodes)++;
    if (n->push_buffer.used) {
        (*leaves)++;
        *bytes += n->push_buffer.used;
        if (n->push_buffer.used > *max_leaf) *max_leaf = (u32)n->push_buffer.used;
    }
    if (n->first_child)
        for (u32 c = 0; c < 8; ++c) count_nodes(n->first_child + c, nodes, leaves, bytes, max_leaf);
}

/* scene-dump format: see tests/ref_io.py:read_scene_dump */
static int
cmd_scene_dump(int argc, char **argv)
{
    if (argc < 7) return 1;
    assemble_scene(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]));
    RefScene *s = g_scene;
    FILE *f = fopen(argv[6], "wb");
    put_u32(f, 0x4e43534fu); /* 'OSCN' */
    put_u32(f, s->parsed.mat_count);
    put_u32(f, s->parsed.sphere_count);
    put_u32(f, s->parsed.box_count);
    put_u32(f, s->parsed.cylinder_count);
    put_u32(f, s->mesh_count);
    put_u32(f, s->parsed.light_count);
    put_u32(f, (u32)s->width);
    put_u32(f, (u32)s->height);
    put_v3(f, s->parsed.ambient);
    put_v3(f, s->camera.p);
    put_v3(f, s->camera.x_axis);
    put_v3(f, s->camera.y_axis);
    put_v3(f, s->camera.z_axis);
    put_v3(f, s->root->aabb_min);
    put_v3(f, s->root->aabb_max);
    for (u32 i = 0; i < s->parsed.mat_count; ++i) {
        Material *m = s->parsed.materials + i;
        put_v3(f, m->diffuse);
        put_v3(f, m->specular.xyz);
        put_f32(f, m->specular.w);
        put_v3(f, m->transmission);
        put_f32(f, m->ior);
        put_v3(f, m->emit_color);
        put_u32(f, (u32)m->is_light);
    }
    for (u32 i = 0; i < s->parsed.sphere_count; ++i) {
        put_v3(f, s->parsed.spheres[i].center);
        put_f32(f, s->parsed.spheres[i].r);
        put_u32(f, s->parsed.spheres[i].mat_index);
    }
    for (u32 i = 0; i < s->parsed.box_count; ++i) {
        put_v3(f, s->parsed.boxes[i].min);
        put_v3(f, s->parsed.boxes[i].max);
        put_u32(f, s->parsed.boxes[i].mat_index);
    }
    for (u32 i = 0; i < s->parsed.cylinder_count; ++i) {
        put_v3(f, s->parsed.cylinders[i].base);
        put_v3(f, s->parsed.cylinders[i].axis);
        put_f32(f, s->parsed.cylinders[i].r);
        put_u32(f, s->parsed.cylinders[i].mat_index);
    }
    /* light list: (type, index into that type's array) */
    {
        TempMemory *lb = &s->world.light_push_buffer;
        for (size_t at = 0; at < lb->used;) {
            ShapeType t = *(ShapeType *)((u8 *)lb->base + at);
            at += sizeof(ShapeType);
            void *p = *(void **)((u8 *)lb->base + at);
            at += sizeof(void *);
            u32 idx = 0xffffffffu;
            if (t == Shape_Type_Sphere) idx = (u32)((Sphere *)p - s->parsed.spheres);
            else if (t == Shape_Type_Cylinder) idx = (u32)((Cylinder *)p - s->parsed.cylinders);
            put_u32(f, (u32)t);
            put_u32(f, idx);
        }
    }
    for (u32 mi = 0; mi < s->mesh_count; ++mi) {
        Mesh *m = s->meshes + mi;
        put_u32(f, m->vertex_count);
        put_u32(f, m->index_count);
        put_u32(f, m->mat_index);
        put_v3(f, m->aabb_min);
        put_v3(f, m->aabb_max);
        fwrite(m->vertices, 12, m->vertex_count, f);
        fwrite(m->indices, 4, m->index_count, f);
    }
    u32 nodes = 0, leaves = 0, max_leaf = 0;
    u64 bytes = 0;
    count_nodes(s->root, &nodes, &leaves, &bytes, &max_leaf);
    put_u32(f, nodes);
    put_u32(f, leaves);
    put_u32(f, (u32)bytes);
    put_u32(f, max_leaf);
    fclose(f);
    printf("{\"materials\": %u, \"spheres\": %u, \"boxes\": %u, \"cylinders\": %u, \"meshes\": %u, \"lights\": %u, "
           "\"octree_nodes\": %u, \"octree_leaves\": %u, \"record_bytes\": %llu, \"max_leaf_bytes\": %u}\n",
           s->parsed.mat_count, s->parsed.sphere_count, s->parsed.box_count, s->parsed.cylinder_count, s->mesh_count,
           s->parsed.light_count, nodes, leaves, (unsigned long long)bytes, max_leaf);
    return 0;
}

static double
now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * render: every policy is a particular way of CALLING the reference function
 * tiled_raytrace_bvh (ray.cpp:1178) -- the policy lives in the caller, exactly as the
 * per-tile seeding lives in main()'s tile loop (macos_main.mm:602-662).
 *   tile32 : main()'s own schedule.  32x32 grid of tiles of ceil(W/32) x ceil(H/32)
 *            pixels, per-tile series = start_random_series(random_u32(&master)) in
 *            row-major tile order, master = start_random_series(seed).
 *   whole  : one call over the whole image, series = start_random_series(random_u32(&master)).
 *   pixel  : one call per pixel (1x1 rect, full spp), series = job_seed(seed, y*W+x).
 *   chunk  : spp is split into spp/chunk calls per pixel of <chunk> samples each; call
 *            k of pixel i uses series job_seed(seed, k*W*H + i); the image is
 *            (sum over k of call results, in k order) / (spp/chunk).
 *   sample : chunk with chunk = 1.
 * Optional rect restricts which pixels are rendered (others are left 0).
 */
static int
cmd_render(int argc, char **argv)
{
    if (argc < 10) return 1;
    i32 W = atoi(argv[4]), H = atoi(argv[5]);
    u32 spp = (u32)strtoul(argv[6], 0, 10);
    u32 seed = (u32)strtoul(argv[7], 0, 10);
    const char *policy = argv[8];
    const char *out_path = argv[9];
    u32 chunk = (argc > 10) ? (u32)strtoul(argv[10], 0, 10) : 1;
    i32 x0 = 0, y0 = 0, x1 = W, y1 = H;
    if (argc > 14) { x0 = atoi(argv[11]); y0 = atoi(argv[12]); x1 = atoi(argv[13]); y1 = atoi(argv[14]); }
    assemble_scene(argv[2], argv[3], W, H);
    RefScene *s = g_scene;
    const f32 rr = 0.8f; /* macos_main.mm:656 */

    v3 *out = (v3 *)calloc((size_t)W * H, sizeof(v3));
    u64 tested = 0;
    u32 final_state = 0;
    double t0 = now_sec();
    if (!strcmp(policy, "tile32")) {
        RandomSeries master = start_random_series(seed);
        i32 tw = ceil_r32_i32(W / (f32)32), th = ceil_r32_i32(H / (f32)32);
        for (i32 ty = 0; ty < 32; ++ty) {
            for (i32 tx = 0; tx < 32; ++tx) {
                i32 ax = tx * tw, ay = ty * th;
                i32 bx = ax + tw, by = ay + th;
                if (bx > W) bx = W;
                if (by > H) by = H;
                RandomSeries series = start_random_series(random_u32(&master));
                /* clip to the requested rect */
                i32 cx0 = ax > x0 ? ax : x0, cy0 = ay > y0 ? ay : y0;
                i32 cx1 = bx < x1 ? bx : x1, cy1 = by < y1 ? by : y1;
                if (cx0 == ax && cy0 == ay && cx1 == bx && cy1 == by) {
                    tested += tiled_raytrace_bvh(&s->world, &s->camera, s->root, out, W, H, ax, ay, bx, by, &series, spp, rr);
                    final_state = series.next_random;
                }
            }
        }
    } else if (!strcmp(policy, "whole")) {
        RandomSeries master = start_random_series(seed);
        RandomSeries series = start_random_series(random_u32(&master));
        tested += tiled_raytrace_bvh(&s->world, &s->camera, s->root, out, W, H, x0, y0, x1, y1, &series, spp, rr);
        final_state = series.next_random;
    } else if (!strcmp(policy, "pixel")) {
        for (i32 y = y0; y < y1; ++y)
            for (i32 x = x0; x < x1; ++x) {
                RandomSeries series = start_random_series(job_seed(seed, (u32)(y * W + x)));
                tested += tiled_raytrace_bvh(&s->world, &s->camera, s->root, out, W, H, x, y, x + 1, y + 1, &series, spp, rr);
                final_state = series.next_random;
            }
    } else if (!strcmp(policy, "chunk") || !strcmp(policy, "sample")) {
        if (!strcmp(policy, "sample")) chunk = 1;
        if (chunk == 0 || spp % chunk) { fprintf(stderr, "spp must be a multiple of chunk\n"); return 2; }
        u32 nchunks = spp / chunk;
        v3 one = {};
        for (i32 y = y0; y < y1; ++y)
            for (i32 x = x0; x < x1; ++x) {
                v3 acc = {};
                for (u32 k = 0; k < nchunks; ++k) {
                    RandomSeries series = start_random_series(job_seed(seed, k * (u32)(W * H) + (u32)(y * W + x)));
                    /* render into a 1x1 scratch image?  No: pixel coordinates feed the camera
                       ray, so render in place and read the pixel back. */
                    tested += tiled_raytrace_bvh(&s->world, &s->camera, s->root, out, W, H, x, y, x + 1, y + 1, &series, chunk, rr);
                    one = out[y * W + x];
                    acc += one;
                    final_state = series.next_random;
                }
                out[y * W + x] = acc / (f32)nchunks;
            }
    } else {
        fprintf(stderr, "unknown policy %s\n", policy);
        return 2;
    }
    double dt = now_sec() - t0;

    FILE *f = fopen(out_path, "wb");
    fwrite(out, sizeof(v3), (size_t)W * H, f);
    fclose(f);
    double paths = (double)(x1 - x0) * (double)(y1 - y0) * (double)spp;
    printf("{\"width\": %d, \"height\": %d, \"spp\": %u, \"seed\": %u, \"policy\": \"%s\", \"chunk\": %u, "
           "\"shapes_tested\": %llu, \"final_rng\": %u, \"seconds\": %.6f, \"paths\": %.0f, \"mpaths_per_s\": %.6f}\n",
           W, H, spp, seed, policy, chunk, (unsigned long long)tested, final_state, dt, paths, paths / dt * 1e-6);
    return 0;
}

/* rng: n states of xor_shift_32 from <seed>, then the f32 draws of the same stream */
static int
cmd_rng(int argc, char **argv)
{
    if (argc < 5) return 1;
    u32 seed = (u32)strtoul(argv[2], 0, 10);
    u32 n = (u32)strtoul(argv[3], 0, 10);
    FILE *f = fopen(argv[4], "wb");
    RandomSeries a = start_random_series(seed);
    for (u32 i = 0; i < n; ++i) {
        f32 v = random_between_0_1(&a);
        put_u32(f, a.next_random);
        put_f32(f, v);
    }
    /* derived draws */
    RandomSeries b = start_random_series(seed);
    for (u32 i = 0; i < n; ++i) put_f32(f, random_between(&b, 0.0f, 2 * pi_32));
    RandomSeries c = start_random_series(seed);
    for (u32 i = 0; i < n; ++i) put_u32(f, random_between_u32(&c, 0, 12));
    RandomSeries d = start_random_series(seed);
    for (u32 i = 0; i < n; ++i) put_v3(f, random_spherical_coordinate(&d, -pi_32 / 2.0f, pi_32 / 2.0f, 0, 2.0f * pi_32));
    put_u32(f, d.next_random);
    fclose(f);
    return 0;
}

/*
 * unit: records of {u32 op; f32 in[24]} -> {f32 out[8]}.
 *  1 triangle  in: v0 v1 v2 o d                 out: t n.xyz inner
 *  2 sphere    in: c r o d                      out: t n.xyz inner
 *  3 aab       in: min max o d                  out: t n.xyz inner
 *  4 cylinder  in: base axis r o d              out: t n.xyz inner
 *  5 sample_brdf in: seed(bits) N wo rough Kd Ks Kt ior   out: wi.xyz is_transmission rng(bits)
 *  6 pdf_brdf  in: N wi wo rough Kd Ks Kt ior   out: p
 *  7 eval_scattering in: N wi wo Kd Ks Kt ior rough dist  out: f.xyz
 *  8 sample_lobe in: N c phi                    out: v.xyz
 *  9 libm      in: x y                          out: sinf(x) cosf(x) atan2f(y,x) powf(x,y) logf(x)
 * 10 normalize in: v                            out: v.xyz
 * 11 fresnel/ggx/geometry in: Ks l_dot_h N H rough w(3)   out: F.xyz D G(w,N,H)
 */
static v3 in_v3(const f32 *p) { return v3_(p[0], p[1], p[2]); }
static void out_hit(f32 *o, IntersectionTestResult r)
{
    o[0] = r.hit_t; o[1] = r.hit_normal.x; o[2] = r.hit_normal.y; o[3] = r.hit_normal.z; o[4] = (f32)r.inner_hit;
}

static int
cmd_unit(int argc, char **argv)
{
    if (argc < 4) return 1;
    PlatformReadFileResult in = read_whole_file(argv[2]);
    size_t rec = 4 + 24 * 4;
    size_t n = in.size / rec;
    FILE *f = fopen(argv[3], "wb");
    for (size_t i = 0; i < n; ++i) {
        u8 *p = in.memory + i * rec;
        u32 op;
        f32 a[24];
        memcpy(&op, p, 4);
        memcpy(a, p + 4, 96);
        f32 o[8] = {};
        switch (op) {
        case 1: out_hit(o, ray_intersect_with_triangle(in_v3(a), in_v3(a + 3), in_v3(a + 6), in_v3(a + 9), in_v3(a + 12))); break;
        case 2: out_hit(o, ray_intersect_with_sphere(in_v3(a), a[3], in_v3(a + 4), in_v3(a + 7))); break;
        case 3: out_hit(o, ray_intersect_with_aab(in_v3(a), in_v3(a + 3), in_v3(a + 6), in_v3(a + 9))); break;
        case 4: out_hit(o, ray_intersect_with_cylinder(in_v3(a), in_v3(a + 3), a[6], in_v3(a + 7), in_v3(a + 10))); break;
        case 5: {
            u32 seed;
            memcpy(&seed, a, 4);
            RandomSeries s = start_random_series(seed);
            SampleBRDFResult r = sample_brdf(&s, in_v3(a + 1), in_v3(a + 4), a[7], in_v3(a + 8), in_v3(a + 11), in_v3(a + 14), a[17]);
            o[0] = r.wi.x; o[1] = r.wi.y; o[2] = r.wi.z; o[3] = (f32)r.is_transmission;
            memcpy(o + 4, &s.next_random, 4);
        } break;
        case 6: o[0] = pdf_brdf(0, in_v3(a), in_v3(a + 3), in_v3(a + 6), a[9], in_v3(a + 10), in_v3(a + 13), in_v3(a + 16), a[19]); break;
        case 7: {
            v3 r = eval_scattering(in_v3(a), in_v3(a + 3), in_v3(a + 6), in_v3(a + 9), in_v3(a + 12), in_v3(a + 15), a[18], a[19], a[20]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
        } break;
        case 8: {
            v3 r = sample_lobe(in_v3(a), a[3], a[4]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
        } break;
        case 9: o[0] = sinf(a[0]); o[1] = cosf(a[0]); o[2] = atan2f(a[1], a[0]); o[3] = powf(a[0], a[1]); o[4] = logf(a[0]); break;
        case 10: {
            v3 r = normalize(in_v3(a));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
        } break;
        case 11: {
            v3 F = fresnel(in_v3(a), a[3]);
            o[0] = F.x; o[1] = F.y; o[2] = F.z;
            o[3] = ggx_distribution(in_v3(a + 4), in_v3(a + 7), a[10]);
            o[4] = geometry(in_v3(a + 11), in_v3(a + 4), in_v3(a + 7), a[10]);
        } break;
        default: break;
        }
        fwrite(o, 4, 8, f);
    }
    fclose(f);
    return 0;
}

/* raycast: rays.bin = n x {o.xyz d.xyz}; out = n x {t n.xyz mat(u32)} via raycast_top_most_node */
static int
cmd_raycast(int argc, char **argv)
{
    if (argc < 6) return 1;
    assemble_scene(argv[2], argv[3], 64, 64);
    RefScene *s = g_scene;
    PlatformReadFileResult in = read_whole_file(argv[4]);
    size_t n = in.size / 24;
    BVHQueue queue = {};
    queue.size = (u32)megabytes(4);
    queue.base = (u8 *)malloc(queue.size);
    FILE *f = fopen(argv[5], "wb");
    u64 tested = 0;
    for (size_t i = 0; i < n; ++i) {
        f32 a[6];
        memcpy(a, in.memory + 24 * i, 24);
        RaycastBVHResult r = raycast_top_most_node(&queue, s->root, &tested, in_v3(a), in_v3(a + 3));
        put_f32(f, r.hit_t);
        put_v3(f, r.hit_normal);
        put_u32(f, r.hit_mat_index);
    }
    fclose(f);
    printf("{\"rays\": %zu, \"shapes_tested\": %llu}\n", n, (unsigned long long)tested);
    return 0;
}

int
main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s scene-dump|render|unit|rng|raycast ...\n", argv[0]);
        return 1;
    }
    if (!strcmp(argv[1], "scene-dump")) return cmd_scene_dump(argc, argv);
    if (!strcmp(argv[1], "render")) return cmd_render(argc, argv);
    if (!strcmp(argv[1], "unit")) return cmd_unit(argc, argv);
    if (!strcmp(argv[1], "rng")) return cmd_rng(argc, argv);
    if (!strcmp(argv[1], "raycast")) return cmd_raycast(argc, argv);
    fprintf(stderr, "unknown command %s\n", argv[1]);
    return 1;
}
