// mir_step_convex.hip — the instantiations of the fused step kernel that carry the convex narrowphase (sphere / capsule geoms:
// GJK on the cores, MPR, closed-form plane cases; mir_convex.h), and the pair-by-pair debug kernel.  Same source as
// mir_step.hip, compiled WITHOUT -fno-signed-zeros (see the note at the launcher there and the Makefile).
#define MIR_STEP_CONVEX_TU
#include "mir_step.hip"
