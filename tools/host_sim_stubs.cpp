/* developer harness: device entry points are absent in the simulation build */
#include "../offline_raytracer_amd/csrc/ort_scene.h"
namespace ort {
int device_count(int *n, std::string *err) { *n = 0; *err = "host_sim: no device"; return ORT_ERR_NO_DEVICE; }
int device_upload(Scene *, int, std::string *err) { *err = "host_sim: no device"; return ORT_ERR_NO_DEVICE; }
void device_release(Scene *) {}
int device_render(Scene *, const ort_render_params *, const ort_tile_job *, uint32_t, void *, float *, void *, uint32_t *, ort_stats *, std::string *err) { *err = "host_sim: no device"; return ORT_ERR_NO_DEVICE; }
uint64_t render_workspace_bytes(const ort_render_params *) { return 0; }
int device_unit_eval(int, const void *, uint32_t, float *, std::string *err) { *err = "host_sim: no device"; return ORT_ERR_NO_DEVICE; }
}
