// mir_scene.h — the object behind a MirHandle (shared by mir_api.hip and mir_render.hip)
#pragma once
#include <stdint.h>

#include "mir_model.h"

struct MirScene {
  int device;
  int B;
  DevModel hm;      // host copy of the compiled model
  HostConsts hc;
  DevModel* dm;     // device copy
  float *qpos, *qvel, *target, *qacc_ws, *poses;
  int32_t *diag, *fkvalid;
  float* prims;     // render primitives (B, ngeom, 32) f32, allocated by the first mir_render
};

// library-internal helpers implemented in mir_api.hip
int mir_set_error(int code, const char* msg);
int mir_refresh_poses(MirScene* h, void* stream);  // make h->poses match qpos for every env (mode-2 launch)
