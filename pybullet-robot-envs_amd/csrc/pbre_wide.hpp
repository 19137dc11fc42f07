// pbre_wide.hpp -- interface between the C-ABI (pbre_capi.hip) and the 64-lane engine (pbre_wide.hip): robots with more
// than 9 DoF (iCub, 32 DoF) are stepped one env per wavefront by the lane-group core (pbre_core.hpp, Shape64).
#pragma once
#include <cstdint>
#include <string>
#include "../../include/pbre.h"

namespace pbre {

struct WideEngine;    // owns the device buffers, stream and events of one batch

int  wide_create(const pbre_config* cfg, WideEngine** out, std::string& err);
void wide_destroy(WideEngine* w);
const char* wide_error(const WideEngine* w);
void wide_dims(const WideEngine* w, int32_t* obs_dim, int32_t* act_dim, int32_t* num_envs, int32_t* state_floats);
int  wide_reset(WideEngine* w, const uint8_t* mask, float* obs);
int  wide_reset_snapshot(WideEngine* w, const uint8_t* mask, float* obs);
int  wide_step(WideEngine* w, const float* actions, float* out);
int  wide_step_device(WideEngine* w, const float* d_actions, float* d_out, void* stream);
int  wide_sync(WideEngine* w);
int  wide_get_state(WideEngine* w, float* s);
int  wide_set_state(WideEngine* w, const float* s);
int  wide_get_state_cols(WideEngine* w, int32_t first, int32_t count, float* out);
int  wide_observe(WideEngine* w, float* obs);
int  wide_settle(WideEngine* w, int32_t n, int32_t flags);
int  wide_set_motors(WideEngine* w, int32_t n, const int32_t* dofs, const float* targets, double kp, double max_force, double max_vel, const uint8_t* mask);
int  wide_apply_action(WideEngine* w, const float* actions, double max_vel);
int  wide_motor_state(WideEngine* w, float* out, const float* in);
int  wide_set_physics(WideEngine* w, const pbre_physics* p);
int  wide_get_physics(const WideEngine* w, pbre_physics* p);
int  wide_set_object_hull(WideEngine* w, const double* verts, int32_t n_verts);
int  wide_get_sweeps(WideEngine* w, int32_t* sweeps);
int  wide_obs_limits(const WideEngine* w, float* lo, float* hi);
int  wide_timing(const WideEngine* w, double* ms, int32_t n);
int  wide_kernel_info(const WideEngine* w, int32_t* info, int32_t n);

}  // namespace pbre
