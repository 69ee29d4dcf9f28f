/*
 * ort_api.cpp -- the C ABI declared in include/ort.h.  Host-side plumbing only: argument
 * checks, the caller-side seeding policies (code/macos_main.mm:602-662) expressed as job
 * lists, and dispatch to the HIP path in ort_kernels.hip.  There is no CPU render path:
 * every render entry point fails unless the scene is resident on a HIP device.
 */
#include <math.h>
#include <string.h>

#include <new>
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

extern "C" {

const char *ort_last_error(void) { return g_error.c_str(); }
int ort_abi_version(void) { return ORT_ABI_VERSION; }

int ort_scene_parse_scn(const char *text, size_t size, const char *base_dir, ort_scene **out) {
    if (!text || !out) return fail(ORT_ERR_INVALID, "null argument");
    *out = nullptr;
    ort_scene *s = new (std::nothrow) ort_scene();
    if (!s) return fail(ORT_ERR_INVALID, "out of memory");
    std::string err;
    int rc = ort::parse_scn_text(text, size, base_dir, s, &err);
    if (rc != ORT_OK) {
        delete s;
        return fail(rc, err);
    }
    s->reference_csg = true; /* as main() does for every scene it loads (macos_main.mm:322-332) */
    *out = s;
    return ORT_OK;
}

int ort_scene_load_scn(const char *scn_path, const char *base_dir, ort_scene **out) {
    if (!scn_path || !out) return fail(ORT_ERR_INVALID, "null argument");
    std::vector<char> text;
    if (ort::read_file(scn_path, &text) != ORT_OK) return fail(ORT_ERR_IO, std::string("cannot read ") + scn_path);
    return ort_scene_parse_scn(text.data(), text.size(), base_dir, out);
}

int ort_scene_create(const ort_scene_desc *d, ort_scene **out) {
    if (!d || !out) return fail(ORT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (d->material_count == 0) return fail(ORT_ERR_INVALID, "material 0 (the reserved no-hit material) is required");
    ort_scene *s = new (std::nothrow) ort_scene();
    if (!s) return fail(ORT_ERR_INVALID, "out of memory");
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

int ort_scene_get_info(const ort_scene *s, ort_scene_info *out) {
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

int ort_scene_get_materials(const ort_scene *s, ort_material *out, uint32_t cap) { return s ? copy_out(s->materials, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
int ort_scene_get_spheres(const ort_scene *s, ort_sphere *out, uint32_t cap) { return s ? copy_out(s->spheres, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
int ort_scene_get_boxes(const ort_scene *s, ort_box *out, uint32_t cap) { return s ? copy_out(s->boxes, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
int ort_scene_get_cylinders(const ort_scene *s, ort_cylinder *out, uint32_t cap) { return s ? copy_out(s->cylinders, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }
int ort_scene_get_lights(const ort_scene *s, ort_light *out, uint32_t cap) { return s ? copy_out(s->lights, out, cap) : fail(ORT_ERR_INVALID, "null scene"); }

int ort_scene_get_mesh(const ort_scene *s, uint32_t i, ort_mesh *out) {
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

int ort_scene_get_camera(const ort_scene *s, int32_t width, int32_t height, ort_camera *out) {
    if (!s || !out || width <= 0 || height <= 0) return fail(ORT_ERR_INVALID, "bad argument");
    ort::camera_basis(*s, width, height, out);
    return ORT_OK;
}

int ort_scene_commit(ort_scene *s) {
    if (!s) return fail(ORT_ERR_INVALID, "null scene");
    std::string err;
    int rc = ort::build_tree(s, &err);
    if (rc == ORT_OK) rc = ort::build_ref_tree(s, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

int ort_scene_get_tree_info(const ort_scene *s, ort_tree_info *out) {
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
    return ORT_OK;
}

int ort_device_count(int *count) {
    if (!count) return fail(ORT_ERR_INVALID, "null argument");
    std::string err;
    int rc = ort::device_count(count, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

int ort_scene_upload(ort_scene *s, int device) {
    if (!s) return fail(ORT_ERR_INVALID, "null scene");
    if (!s->tree.built) return fail(ORT_ERR_STATE, "ort_scene_commit has not been called");
    std::string err;
    int rc = ort::device_upload(s, device, &err);
    return rc == ORT_OK ? ORT_OK : fail(rc, err);
}

int ort_tiled_raytrace_batch(ort_scene *s, float *out_rgb, int32_t width, int32_t height, const ort_tile_job *jobs,
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

int ort_tiled_raytrace(ort_scene *s, float *out_rgb, int32_t width, int32_t height, int32_t x0, int32_t y0, int32_t x1,
                       int32_t y1, uint32_t *rng_state, uint32_t spp, float rr, uint64_t *shape_tests) {
    if (!rng_state) return fail(ORT_ERR_INVALID, "null rng_state");
    ort_tile_job j{x0, y0, x1, y1, *rng_state, spp};
    uint32_t final_state = *rng_state;
    ort_stats st{};
    int rc = ort_tiled_raytrace_batch(s, out_rgb, width, height, &j, 1, rr, &final_state, &st);
    if (rc != ORT_OK) return rc;
    *rng_state = final_state;
    if (shape_tests) *shape_tests = st.tri_tests + st.analytic_tests;
    return ORT_OK;
}

int ort_render_image(ort_scene *s, const ort_render_params *p, float *out_rgb, ort_stats *stats) {
    if (!out_rgb) return fail(ORT_ERR_INVALID, "null framebuffer");
    return render_common(s, p, nullptr, out_rgb, nullptr, stats);
}

int ort_render_image_device(ort_scene *s, const ort_render_params *p, void *d_out_rgb, void *hip_stream, ort_stats *stats) {
    if (!d_out_rgb) return fail(ORT_ERR_INVALID, "null device framebuffer");
    return render_common(s, p, d_out_rgb, nullptr, hip_stream, stats);
}

int ort_unit_eval_device(int device, const void *records, uint32_t count, float *out) {
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

int ort_render_workspace_bytes(const ort_render_params *p, uint64_t *bytes) {
    if (!p || !bytes) return fail(ORT_ERR_INVALID, "null argument");
    *bytes = ort::render_workspace_bytes(p);
    return ORT_OK;
}

} // extern "C"
