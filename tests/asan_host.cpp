// Sanitizer driver for the host-compiled product code (tests/test_sanitizers_cpu.py; built by `make -C gym-genesis_amd/csrc asan-host`
// together with mir_compile.cpp / mir_compile64.cpp): one C entry point that runs a MirSceneSpec through both scene compilers and
// the spec emitter, and three stand-ins with the signatures of mir_step_prepare / mir_step_go / mir_step_end for the _mirfast module.
// Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <new>

#include "mir_model.h"
#include "mir_model64.h"

extern "C" int mir_debug_emit_spec(const MirSceneSpec* spec, const char* name, char* out, int32_t cap);

extern "C" {

// returns rc16 | rc64 << 8 | (emit ok) << 16 (the return codes as small non-negative numbers: -rc)
int asan_compile_spec(const MirSceneSpec* spec, char* err16, char* err64) {
  DevModel* m = new DevModel;
  DevModel64* m64 = new DevModel64;
  HostConsts hc;
  memset(m, 0, sizeof *m);
  memset(m64, 0, sizeof *m64);
  const int rc16 = mir_compile_model(spec, m, &hc, err16);
  const int rc64 = mir_compile_model64(spec, m64, &hc, err64);
  int emit = 0;
  if (rc16 == 0) {
    char* text = new char[1 << 16];
    emit = mir_debug_emit_spec(spec, "SpecProbe", text, 1 << 16) > 0 ? 1 : 0;
    delete[] text;
  }
  delete m;
  delete m64;
  return (-rc16) | (-rc64) << 8 | emit << 16;
}

static int g_calls[3];
int asan_stub_prepare(void* h, float* a, float* b, float* c, uint8_t* d) { g_calls[0]++; return h && a && b && c && d ? 0 : -1; }
int asan_stub_go(void* h, const float* action, void* stream) { g_calls[1]++; return h && action ? 0 : -1; }
int asan_stub_end(void* h, uint8_t* host) { g_calls[2]++; if (host) memset(host, 1, 8); return 0; }
int asan_stub_calls(int k) { return g_calls[k]; }
}
