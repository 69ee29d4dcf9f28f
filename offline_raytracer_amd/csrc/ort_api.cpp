/*
 * ort_api.cpp -- the C ABI declared in include/ort.h.  Host-side plumbing only: argument
 * checks, the caller-side seeding policies (code/macos_main.mm:602-662) expressed as job
 * lists, and dispatch to the HIP path in ort_kernels.hip.  There is no CPU render path:
 * every render entry point fails unless the scene is resident on a HIP device.
 */
#include <math.h>
#include <string.h>

#include <exception>
#include <new>
#include <memory>
#include <string>
#include <vector>

#include "ort_scene.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

/* random_u32 (random.h:83-89) on the master series: used for the per-tile seeds */
uint32_t xorshift(uint32_t *s) {
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x >> 5; /* sic: third shift is right (random.h:10-12) */
    *s = x;
    return x;
}

int check_params(const ort_scene *scene, const ort_render_params *p) {
    if (!scene || !p) return fail(ORT_ERR_INVALID, "null scene or params");
    if (p->width <= 0 || p->height <= 0) return fail(ORT_ERR_INVALID, "image size must be positive");
    if ((uint64_t)p->width * (uint64_t)p->height > 0x7fffffffull / 3) return fail(ORT_ERR_INVALID, "image too large");
    if (p->width > 65535 || p->height > 65535) return fail(ORT_ERR_UNSUPPORTED, "image side above 65535 (lane state packs coordinates in 16 bits)");
    if (p->policy == ORT_POLICY_CHUNK && p->chunk && p->spp / p->chunk > 65535u) return fail(ORT_ERR_UNSUPPORTED, "more than 65535 chunks");
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 >= p->x1 || p->y0 >= p->y1)
        return fail(ORT_ERR_INVALID, "render rect is empty or outside the image");
    if (p->spp == 0) return fail(ORT_ERR_INVALID, "spp must be >= 1");
    if (!(p->rr >= 0.0f)) return fail(ORT_ERR_INVALID, "rr must be >= 0");
    if (p->policy == ORT_POLICY_CHUNK && (p->chunk == 0 || p->spp % p->chunk))
        return fail(ORT_ERR_INVALID, "spp must be a multiple of chunk");
    if (p->policy < ORT_POLICY_TILE32 || p->policy > ORT_POLICY_CHUNK) return fail(ORT_ERR_INVALID, "unknown policy");
    if (p->shard_count > 1 && p->shard_index >= p->shard_count) return fail(ORT_ERR_INVALID, "shard index out of range");
    if (p->shard_count > 1 && (p->policy == ORT_POLICY_TILE32 || p->policy == ORT_POLICY_WHOLE))
        return fail(ORT_ERR_UNSUPPORTED, "sharding needs a per-pixel seeding policy (PIXEL or CHUNK)");
    if (p->flags & ORT_RENDER_PACKED) {
        if (p->policy == ORT_POLICY_TILE32 || p->policy == ORT_POLICY_WHOLE)
            return fail(ORT_ERR_UNSUPPORTED, "a packed framebuffer needs a per-pixel seeding policy (PIXEL or CHUNK)");
        if (p->x0 != 0 || p->y0 != 0 || p->x1 != p->width || p->y1 != p->height)
            return fail(ORT_ERR_INVALID, "a packed framebuffer covers the whole image: the rect must be the full frame");
    }
    if (!scene->tree.built) return fail(ORT_ERR_STATE, "ort_scene_commit has not been called");
    if (!scene->dev) return fail(ORT_ERR_NO_DEVICE, "scene is not resident on a HIP device: call ort_scene_upload (no CPU fallback)");
    return ORT_OK;
}

/* main()'s tile schedule (macos_main.mm:602-662) as explicit jobs: every one of the 1024
   tiles draws its seed, rendered or not; only tiles wholly inside the rect are rendered */
void tile32_jobs(const ort_render_params *p, std::vector<ort_tile_job> *jobs) {
    uint32_t master = p->seed;
    int32_t tw = (int32_t)ceilf(p->width / (float)32), th = (int32_t)ceilf(p->height / (float)32);
    for (int32_t ty = 0; ty < 32; ++ty)
        for (int32_t tx = 0; tx < 32; ++tx) {
            ort_tile_job j;
            j.x0 = tx * tw; j.y0 = ty * th;
            j.x1 = j.x0 + tw; j.y1 = j.y0 + th;
            if (j.x1 > p->width) j.x1 = p->width;
            if (j.y1 > p->height) j.y1 = p->height;
            j.rng_state = xorshift(&master);
            j.spp = p->spp;
            if (j.x0 >= p->x0 && j.y0 >= p->y0 && j.x1 <= p->x1 && j.y1 <= p->y1 && j.x0 < j.x1 && j.y0 < j.y1)
                jobs->push_back(j);
        }
}

int render_common(ort_scene *scene, const ort_render_params *p, void *d_out, float *h_out, void *stream, ort_stats *stats) {
    int rc = check_params(scene, p);
    if (rc != ORT_OK) return rc;
    std::string err;
    if (p->policy == ORT_POLICY_TILE32 || p->policy == ORT_POLICY_WHOLE) {
        std::vector<ort_tile_job> jobs;
        if (p->policy == ORT_POLICY_TILE32) {
            tile32_jobs(p, &jobs);
        } else {
            uint32_t master = p->seed;
            ort_tile_job j{p->x0, p->y0, p->x1, p->y1, xorshift(&master), p->spp};
            jobs.push_back(j);
        }
        if (jobs.empty()) return ORT_OK;
        rc = ort::device_render(scene, p, jobs.data(), (uint32_t)jobs.size(), d_out, h_out, stream, nullptr, stats, &err);
    } else {
        rc = ort::device_render(scene, p, nullptr, 0, d_out, h_out, stream, nullptr, stats, &err);
    }
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

template <typename T>
int copy_out(const std::vector<T> &v, T *out, uint32_t cap) {
    if (v.size() > cap || (!out && !v.empty())) return fail(ORT_ERR_INVALID, "output capacity too small");
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(T));
    return ORT_OK;
}

} // namespace


/* No exception crosses the C boundary: an allocation failure on hostile input (or anything else the C++ side
   throws) becomes an error code with a message instead of std::terminate under the caller's feet. */
template <typename F>
int guarded(F f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(ORT_ERR_NO_MEMORY, "out of memory");
    } catch (const std::exception &e) {
        return fail(ORT_ERR_INTERNAL, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(ORT_ERR_INTERNAL, "internal error");
    }
}

extern "C" {

const char *ort_last_error(void) { return g_error.c_str(); }
static int ort_abi_version_impl(void) { return ORT_ABI_VERSION; }

static int ort_scene_parse_scn_impl(const char *text, size_t size, const char *base_dir, ort_scene **out) {
    if (!text || !out) return fail(ORT_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<ort_scene> s(new (std::nothrow) ort_scene()); /* owned here until handed out: the parser may throw (bad_alloc on hostile input) */
    if (!s) return fail(ORT_ERR_NO_MEMORY, "out of memory");
    std::string err;
    int rc = ort::parse_scn_text(text, size, base_dir, s.get(), &err);
    if (rc != ORT_OK) return fail(rc, err);
    s->reference_csg = true; /* as main() does for every scene it loads (macos_main.mm:322-332) */
    *out = s.release();
    return ORT_OK;
}

static int ort_scene_load_scn_impl(const char *scn_path, const char *base_dir, ort_scene **out) {
    if (!scn_path || !out) return fail(ORT_ERR_INVALID, "null argument");
    std::vector<char> text;
    if (ort::read_file(scn_path, &text) != ORT_OK) return fail(ORT_ERR_IO, std::string("cannot read ") + scn_path);
    return ort_scene_parse_scn_impl(text.data(), text.size(), base_dir, out);
}

static int ort_scene_create_impl(const ort_scene_desc *d, ort_scene **out) {
    if (!d || !out) return fail(ORT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (d->material_count == 0) return fail(ORT_ERR_INVALID, "material 0 (the reserved no-hit material) is required");
    ort_scene *s = new (std::nothrow) ort_scene();
    if (!s) return fail(ORT_ERR_NO_MEMORY, "out of memory");
    s->materials.assign(d->materials, d->materials + d->material_count);
    if (d->sphere_count) s->spheres.assign(d->spheres, d->spheres + d->sphere_count);
    if (d->box_count) s->boxes.assign(d->boxes, d->boxes + d->box_count);
    if (d->cylinder_count) s->cylinders.assign(d->cylinders, d->cylinders + d->cylinder_count);
    if (d->light_count) s->lights.assign(d->lights, d->lights + d->light_count);
    auto bad_mat = [&](uint32_t m) { return m >= d->material_count; };
    bool bad = false;
    for (auto &x : s->spheres) bad |= bad_mat(x.mat);
    for (auto &x : s->boxes) bad |= bad_mat(x.mat);
    for (auto &x : s->cylinders) bad |= bad_mat(x.mat);
    for (auto &l : s->lights) {
        if (l.type == 1u) bad |= (l.index >= d->sphere_count);
        else if (l.type == 2u) bad |= (l.index >= d->cylinder_count);
        else bad = true;
    }
    for (uint32_t i = 0; i < d->mesh_count && !bad; ++i) {
        const ort_mesh &m = d->meshes[i];
        ort::HostMesh hm;
        hm.vertices.assign(m.vertices, m.vertices + 3 * (size_t)m.vertex_count);
        hm.indices.assign(m.indices, m.indices + m.index_count);
        hm.mat = m.mat;
        hm.aabb_min = m.aabb_min;
        hm.aabb_max = m.aabb_max;
        bad |= bad_mat(m.mat);
        for (uint32_t ix : hm.indices) bad |= (ix >= m.vertex_count);
        s->meshes.push_back(std::move(hm));
    }
    if (bad) {
        delete s;
        return fail(ORT_ERR_INVALID, "material, light or vertex index out of range");
    }
    s->ambient = d->ambient;
    s->camera_p = d->camera_p;
    memcpy(s->camera_quat, d->camera_quat_xyzw, sizeof(s->camera_quat));
    s->camera_height_ratio = d->camera_height_ratio;
    s->screen_width = d->screen_width;
    s->screen_height = d->screen_height;
    s->reference_csg = d->with_reference_csg != 0;
    *out = s;
    return ORT_OK;
}

void ort_scene_destroy(ort_scene *scene) {
    if (!scene) return;
    ort::device_release(scene);
    delete scene;
}

static int ort_scene_get_info_impl(const ort_scene *s, ort_scene_info *out) {
    if (!s || !out) return fail(ORT_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    out->material_count = (uint32_t)s->materials.size();
    out->sphere_count = (uint32_t)s->spheres.size();
    out->box_count = (uint32_t)s->boxes.size();
    out->cylinder_count = (uint32_t)s->cylinders.size();
    out->mesh_count = (uint32_t)s->meshes.size();
    out->light_count = (uint32_t)s->lights.size();
    for (const auto &m : s->meshes) out->triangle_count += (uint32_t)(m.indices.size() / 3);
    out->screen_width = s->screen_width;
    out->screen_height = s->screen_height;
    out->ambient = s->ambient;
    out->camera_p = s->camera_p;
    memcpy(out->camera_quat_xyzw, s->camera_quat, sizeof(s->camera_quat));
    out->camera_height_ratio = s->camera_height_ratio;
    return ORT_OK;
}

static int ort_scene_get_materials_impl(const ort_scene *s, ort_material *out, uint32_t cap) { return s ? copy_out(s->materials, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
static int ort_scene_get_spheres_impl(const ort_scene *s, ort_sphere *out, uint32_t cap) { return s ? copy_out(s->spheres, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
static int ort_scene_get_boxes_impl(const ort_scene *s, ort_box *out, uint32_t cap) { return s ? copy_out(s->boxes, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
static int ort_scene_get_cylinders_impl(const ort_scene *s, ort_cylinder *out, uint32_t cap) { return s ? copy_out(s->cylinders, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
static int ort_scene_get_lights_impl(const ort_scene *s, ort_light *out, uint32_t cap) { return s ? copy_out(s->lights, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }

static int ort_scene_get_mesh_impl(const ort_scene *s, uint32_t i, ort_mesh *out) {
    if (!s || !out) return fail(ORT_ERR_INVALID, "null argument");
    if (i >= s->meshes.size()) return fail(ORT_ERR_INVALID, "mesh index out of range");
    const ort::HostMesh &m = s->meshes[i];
    out->vertices = m.vertices.data();
    out->vertex_count = (uint32_t)(m.vertices.size() / 3);
    out->indices = m.indices.data();
    out->index_count = (uint32_t)m.indices.size();
    out->mat = m.mat;
    out->aabb_min = m.aabb_min;
    out->aabb_max = m.aabb_max;
    return ORT_OK;
}

static int ort_scene_get_camera_impl(const ort_scene *s, int32_t width, int32_t height, ort_camera *out) {
    if (!s || !out || width <= 0 || height <= 0) return fail(ORT_ERR_INVALID, "bad argument");
    ort::camera_basis(*s, width, height, out);
    return ORT_OK;
}

static int ort_scene_commit_impl(ort_scene *s) {
    if (!s) return fail(ORT_ERR_INVALID, "null scene");
    std::string err;
    int rc = ort::build_tree(s, &err);
    if (rc == ORT_OK) rc = ort::build_ref_tree(s, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

static int ort_scene_get_tree_info_impl(const ort_scene *s, ort_tree_info *out) {
    if (!s || !out) return fail(ORT_ERR_INVALID, "null argument");
    if (!s->tree.built) return fail(ORT_ERR_STATE, "ort_scene_commit has not been called");
    const ort::Tree &t = s->tree;
    memset(out, 0, sizeof(*out));
    out->node_count = (uint32_t)t.nodes.size();
    out->leaf_count = t.leaf_count;
    out->max_leaf_prims = t.max_leaf_prims;
    out->max_depth = t.max_depth;
    out->node_bytes = t.nodes.size() * sizeof(ort::DevNode);
    out->prim_bytes = t.tris.size() * (sizeof(ort::DevTri) + 4) + t.spheres.size() * (sizeof(ort::DevSphere) + 4) +
                      t.boxes.size() * (sizeof(ort::DevBox) + 4) + t.cyls.size() * (sizeof(ort::DevCyl) + 4);
    out->sah_cost = t.sah_cost;
    const ort::RefTree &r = s->ref;
    out->ref_node_count = (uint32_t)r.nodes.size();
    out->ref_nonempty_leaves = r.nonempty_leaves;
    out->ref_max_leaf_records = r.max_leaf_records;
    out->ref_bytes = r.nodes.size() * sizeof(ort::DevRefNode) + r.recs.size() * 4 + r.chain_boxes.size() * 16 +
                     (r.tri_chain.size() + r.sphere_chain.size() + r.box_chain.size() + r.cyl_chain.size()) * 8;
    out->prologue_prims = t.pro_boxes + t.pro_spheres + t.pro_cyls;
    out->wide_node_count = (uint32_t)t.nodes4.size();
    out->wide_max_depth = t.max_depth4;
    return ORT_OK;
}

static int ort_device_count_impl(int *count) {
    if (!count) return fail(ORT_ERR_INVALID, "null argument");
    std::string err;
    int rc = ort::device_count(count, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

static int ort_scene_upload_impl(ort_scene *s, int device) {
    if (!s) return fail(ORT_ERR_INVALID, "null scene");
    if (!s->tree.built) return fail(ORT_ERR_STATE, "ort_scene_commit has not been called");
    std::string err;
    int rc = ort::device_upload(s, device, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

static int ort_tiled_raytrace_batch_impl(ort_scene *s, float *out_rgb, int32_t width, int32_t height, const ort_tile_job *jobs,
                             uint32_t job_count, float rr, uint32_t *final_states, ort_stats *stats) {
    if (!s || !out_rgb || (!jobs && job_count)) return fail(ORT_ERR_INVALID, "null argument");
    ort_render_params p{};
    p.width = width; p.height = height;
    p.x0 = 0; p.y0 = 0; p.x1 = width; p.y1 = height;
    p.policy = ORT_POLICY_WHOLE;
    p.spp = 1; p.rr = rr;
    if (stats) p.flags = ORT_RENDER_COUNTERS;
    int rc = check_params(s, &p);
    if (rc != ORT_OK) return rc;
    for (uint32_t i = 0; i < job_count; ++i) {
        const ort_tile_job &j = jobs[i];
        if (j.x0 < 0 || j.y0 < 0 || j.x1 > width || j.y1 > height) return fail(ORT_ERR_INVALID, "job rect outside the image");
        if (j.spp == 0) return fail(ORT_ERR_INVALID, "job spp must be >= 1");
    }
    if (job_count == 0) return ORT_OK;
    std::string err;
    rc = ort::device_render(s, &p, jobs, job_count, nullptr, out_rgb, nullptr, final_states, stats, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

static int ort_tiled_raytrace_impl(ort_scene *s, float *out_rgb, int32_t width, int32_t height, int32_t x0, int32_t y0, int32_t x1,
                       int32_t y1, uint32_t *rng_state, uint32_t spp, float rr, uint64_t *shape_tests) {
    if (!rng_state) return fail(ORT_ERR_INVALID, "null rng_state");
    ort_tile_job j{x0, y0, x1, y1, *rng_state, spp};
    uint32_t final_state = *rng_state;
    ort_stats st{};
    int rc = ort_tiled_raytrace_batch_impl(s, out_rgb, width, height, &j, 1, rr, &final_state, &st);
    if (rc != ORT_OK) return rc;
    *rng_state = final_state;
    if (shape_tests) *shape_tests = st.tri_tests + st.analytic_tests;
    return ORT_OK;
}

static int ort_render_image_impl(ort_scene *s, const ort_render_params *p, float *out_rgb, ort_stats *stats) {
    if (!out_rgb) return fail(ORT_ERR_INVALID, "null framebuffer");
    return render_common(s, p, nullptr, out_rgb, nullptr, stats);
}

static int ort_render_image_device_impl(ort_scene *s, const ort_render_params *p, void *d_out_rgb, void *hip_stream, ort_stats *stats) {
    if (!d_out_rgb) return fail(ORT_ERR_INVALID, "null device framebuffer");
    return render_common(s, p, d_out_rgb, nullptr, hip_stream, stats);
}

static int ort_unit_eval_device_impl(int device, const void *records, uint32_t count, float *out) {
    if ((!records || !out) && count) return fail(ORT_ERR_INVALID, "null argument");
    /* op 4 (cylinder): the kernel consumes the host-precomputed frame; fill it in a copy */
    std::vector<unsigned char> copy((const unsigned char *)records, (const unsigned char *)records + (size_t)count * 100u);
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t op;
        float a[24];
        memcpy(&op, &copy[(size_t)i * 100u], 4);
        if (op != 4u) continue;
        memcpy(a, &copy[(size_t)i * 100u + 4], 96);
        ort_cylinder c{{a[0], a[1], a[2]}, {a[3], a[4], a[5]}, a[6], 0};
        ort::cylinder_frame_for(c, a + 13, a + 22);
        memcpy(&copy[(size_t)i * 100u + 4], a, 96);
    }
    std::string err;
    int rc = ort::device_unit_eval(device, copy.data(), count, out, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

static int ort_render_workspace_bytes_impl(const ort_render_params *p, uint64_t *bytes) {
    if (!p || !bytes) return fail(ORT_ERR_INVALID, "null argument");
    *bytes = ort::render_workspace_bytes(p);
    return ORT_OK;
}


/* ---- multi-GPU (ort_comm.cpp) ---- */
static int ort_shard_block_count_impl(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, uint64_t *blocks) {
    if (!blocks || width <= 0 || height <= 0 || (shard_count > 1 && shard_index >= shard_count)) return fail(ORT_ERR_INVALID, "bad argument");
    *blocks = ort::comm_shard_blocks(width, height, shard_count > 1 ? shard_index : 0, shard_count > 1 ? shard_count : 1);
    return ORT_OK;
}
static int check_shard(const void *a, const void *b, int32_t w, int32_t h, uint32_t index, uint32_t count) {
    if (!a || !b || w <= 0 || h <= 0 || count == 0 || index >= count) return fail(ORT_ERR_INVALID, "bad argument");
    return ORT_OK;
}
static int ort_pack_blocks_host_impl(const float *full_rgb, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *packed) {
    int rc = check_shard(full_rgb, packed, width, height, shard_index, shard_count);
    if (rc == ORT_OK) ort::pack_blocks_host(full_rgb, width, height, shard_index, shard_count, packed);
    return rc;
}
static int ort_unpack_blocks_host_impl(const float *packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *full_rgb) {
    int rc = check_shard(full_rgb, packed, width, height, shard_index, shard_count);
    if (rc == ORT_OK) ort::unpack_blocks_host(packed, width, height, shard_index, shard_count, full_rgb);
    return rc;
}
static int ort_unpack_blocks_device_impl(const void *d_packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count,
                                         void *d_full_rgb, void *hip_stream) {
    int rc = check_shard(d_full_rgb, d_packed, width, height, shard_index, shard_count);
    if (rc != ORT_OK) return rc;
    std::string err;
    rc = ort::unpack_blocks_device(d_packed, width, height, shard_index, shard_count, d_full_rgb, hip_stream, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}
static int ort_comm_unique_id_impl(void *id) {
    if (!id) return fail(ORT_ERR_INVALID, "null argument");
    std::string err;
    int rc = ort::comm_unique_id(id, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}
static int ort_comm_create_impl(const void *id, int rank, int world, int device, ort_comm **out) {
    if (!out) return fail(ORT_ERR_INVALID, "null argument");
    std::string err;
    ort::Comm *c = nullptr;
    int rc = ort::comm_create(id, rank, world, device, &c, &err);
    *out = (ort_comm *)c;
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}
static int ort_comm_create_local_impl(int world, const int *devices, ort_comm **out) {
    if (!out || world < 1) return fail(ORT_ERR_INVALID, "bad argument");
    std::string err;
    int rc = ort::comm_create_local(world, devices, (ort::Comm **)out, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}
static int ort_gather_framebuffer_impl(ort_comm *comm, const void *d_packed, void *d_full_rgb, int32_t width, int32_t height, void *hip_stream) {
    if (!comm || !d_packed || width <= 0 || height <= 0) return fail(ORT_ERR_INVALID, "bad argument");
    std::string err;
    int rc = ort::gather_framebuffer((ort::Comm *)comm, d_packed, d_full_rgb, width, height, hip_stream, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}
static int ort_gather_framebuffer_local_impl(ort_comm **comms, int world, const void *const *d_packed, void *d_full_rgb_rank0, int32_t width,
                                             int32_t height, void *const *hip_streams) {
    if (!comms || !d_packed || !d_full_rgb_rank0 || world < 1 || width <= 0 || height <= 0) return fail(ORT_ERR_INVALID, "bad argument");
    std::string err;
    int rc = ort::gather_framebuffer_local((ort::Comm **)comms, world, d_packed, d_full_rgb_rank0, width, height, hip_streams, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

/* ---- the exported entry points: every one behind guarded() ---- */
int ort_abi_version(void) { return guarded([&]() { return ort_abi_version_impl(); }); }
int ort_scene_parse_scn(const char *text, size_t size, const char *base_dir, ort_scene **out) { return guarded([&]() { return ort_scene_parse_scn_impl(text, size, base_dir, out); }); }
int ort_scene_load_scn(const char *scn_path, const char *base_dir, ort_scene **out) { return guarded([&]() { return ort_scene_load_scn_impl(scn_path, base_dir, out); }); }
int ort_scene_create(const ort_scene_desc *d, ort_scene **out) { return guarded([&]() { return ort_scene_create_impl(d, out); }); }
int ort_scene_get_info(const ort_scene *s, ort_scene_info *out) { return guarded([&]() { return ort_scene_get_info_impl(s, out); }); }
int ort_scene_get_materials(const ort_scene *s, ort_material *out, uint32_t cap) { return guarded([&]() { return ort_scene_get_materials_impl(s, out, cap); }); }
int ort_scene_get_spheres(const ort_scene *s, ort_sphere *out, uint32_t cap) { return guarded([&]() { return ort_scene_get_spheres_impl(s, out, cap); }); }
int ort_scene_get_boxes(const ort_scene *s, ort_box *out, uint32_t cap) { return guarded([&]() { return ort_scene_get_boxes_impl(s, out, cap); }); }
int ort_scene_get_cylinders(const ort_scene *s, ort_cylinder *out, uint32_t cap) { return guarded([&]() { return ort_scene_get_cylinders_impl(s, out, cap); }); }
int ort_scene_get_lights(const ort_scene *s, ort_light *out, uint32_t cap) { return guarded([&]() { return ort_scene_get_lights_impl(s, out, cap); }); }
int ort_scene_get_mesh(const ort_scene *s, uint32_t i, ort_mesh *out) { return guarded([&]() { return ort_scene_get_mesh_impl(s, i, out); }); }
int ort_scene_get_camera(const ort_scene *s, int32_t width, int32_t height, ort_camera *out) { return guarded([&]() { return ort_scene_get_camera_impl(s, width, height, out); }); }
int ort_scene_commit(ort_scene *s) { return guarded([&]() { return ort_scene_commit_impl(s); }); }
int ort_scene_get_tree_info(const ort_scene *s, ort_tree_info *out) { return guarded([&]() { return ort_scene_get_tree_info_impl(s, out); }); }
int ort_device_count(int *count) { return guarded([&]() { return ort_device_count_impl(count); }); }
int ort_scene_upload(ort_scene *s, int device) { return guarded([&]() { return ort_scene_upload_impl(s, device); }); }
int ort_tiled_raytrace_batch(ort_scene *s, float *out_rgb, int32_t width, int32_t height, const ort_tile_job *jobs, uint32_t job_count, float rr, uint32_t *final_states, ort_stats *stats) { return guarded([&]() { return ort_tiled_raytrace_batch_impl(s, out_rgb, width, height, jobs, job_count, rr, final_states, stats); }); }
int ort_tiled_raytrace(ort_scene *s, float *out_rgb, int32_t width, int32_t height, int32_t x0, int32_t y0, int32_t x1, int32_t y1, uint32_t *rng_state, uint32_t spp, float rr, uint64_t *shape_tests) { return guarded([&]() { return ort_tiled_raytrace_impl(s, out_rgb, width, height, x0, y0, x1, y1, rng_state, spp, rr, shape_tests); }); }
int ort_render_image(ort_scene *s, const ort_render_params *p, float *out_rgb, ort_stats *stats) { return guarded([&]() { return ort_render_image_impl(s, p, out_rgb, stats); }); }
int ort_render_image_device(ort_scene *s, const ort_render_params *p, void *d_out_rgb, void *hip_stream, ort_stats *stats) { return guarded([&]() { return ort_render_image_device_impl(s, p, d_out_rgb, hip_stream, stats); }); }
int ort_unit_eval_device(int device, const void *records, uint32_t count, float *out) { return guarded([&]() { return ort_unit_eval_device_impl(device, records, count, out); }); }
int ort_render_workspace_bytes(const ort_render_params *p, uint64_t *bytes) { return guarded([&]() { return ort_render_workspace_bytes_impl(p, bytes); }); }
int ort_shard_block_count(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, uint64_t *blocks) { return guarded([&]() { return ort_shard_block_count_impl(width, height, shard_index, shard_count, blocks); }); }
int ort_pack_blocks_host(const float *full_rgb, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *packed) { return guarded([&]() { return ort_pack_blocks_host_impl(full_rgb, width, height, shard_index, shard_count, packed); }); }
int ort_unpack_blocks_host(const float *packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, float *full_rgb) { return guarded([&]() { return ort_unpack_blocks_host_impl(packed, width, height, shard_index, shard_count, full_rgb); }); }
int ort_unpack_blocks_device(const void *d_packed, int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count, void *d_full_rgb, void *hip_stream) { return guarded([&]() { return ort_unpack_blocks_device_impl(d_packed, width, height, shard_index, shard_count, d_full_rgb, hip_stream); }); }
int ort_comm_unique_id(void *id) { return guarded([&]() { return ort_comm_unique_id_impl(id); }); }
int ort_comm_create(const void *id, int rank, int world, int device, ort_comm **out) { return guarded([&]() { return ort_comm_create_impl(id, rank, world, device, out); }); }
int ort_comm_create_local(int world, const int *devices, ort_comm **out) { return guarded([&]() { return ort_comm_create_local_impl(world, devices, out); }); }
int ort_gather_framebuffer(ort_comm *comm, const void *d_packed, void *d_full_rgb, int32_t width, int32_t height, void *hip_stream) { return guarded([&]() { return ort_gather_framebuffer_impl(comm, d_packed, d_full_rgb, width, height, hip_stream); }); }
int ort_gather_framebuffer_local(ort_comm **comms, int world, const void *const *d_packed, void *d_full_rgb_rank0, int32_t width, int32_t height, void *const *hip_streams) { return guarded([&]() { return ort_gather_framebuffer_local_impl(comms, world, d_packed, d_full_rgb_rank0, width, height, hip_streams); }); }
void ort_comm_destroy(ort_comm *comm) { try { ort::comm_destroy((ort::Comm *)comm); } catch (...) {} }

} // extern "C"
