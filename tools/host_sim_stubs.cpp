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
/* the multi-GPU entry points of ort_api.cpp (ort_comm.cpp is HIP code): never called by the harness */
namespace ort {
uint64_t shard_block_count(const ort_render_params *) { return 0; }
uint64_t comm_shard_blocks(int32_t, int32_t, uint32_t, uint32_t) { return 0; }
void pack_blocks_host(const float *, int32_t, int32_t, uint32_t, uint32_t, float *) {}
void unpack_blocks_host(const float *, int32_t, int32_t, uint32_t, uint32_t, float *) {}
int unpack_blocks_device(const void *, int32_t, int32_t, uint32_t, uint32_t, void *, void *, std::string *) { return ORT_ERR_NO_DEVICE; }
int comm_unique_id(void *, std::string *) { return ORT_ERR_NO_DEVICE; }
int comm_create(const void *, int, int, int, Comm **, std::string *) { return ORT_ERR_NO_DEVICE; }
int comm_create_local(int, const int *, Comm **, std::string *) { return ORT_ERR_NO_DEVICE; }
void comm_destroy(Comm *) {}
int gather_framebuffer(Comm *, const void *, void *, int32_t, int32_t, void *, std::string *) { return ORT_ERR_NO_DEVICE; }
int gather_framebuffer_local(Comm **, int, const void *const *, void *, int32_t, int32_t, void *const *, std::string *) { return ORT_ERR_NO_DEVICE; }
}
