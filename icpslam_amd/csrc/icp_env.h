// icp_env.h -- environment switches of libicpgpu.so, in two classes.
//
// PRODUCTION switches are read by every build (std::getenv directly; the list in include/icpgpu.h, section "environment",
// and INTEGRATION.md repeat this one):
//   ICPGPU_WAIT_TIMEOUT_MS   deadline of every host wait for the device (mailbox, gather); default 30 000
//   ICPGPU_BATCH_THREADS / ICPGPU_BATCH_DEPTH / ICPGPU_BATCH_GROUPS   icpgpu_align_batch: host threads, pairs per lock-step
//                            group, groups in flight on the GPU (all threads together; default 8)
//   ICPGPU_RECOGNISE=0       icpgpu_set_target / icpgpu_set_source always upload (no content recognition)
//   ICPGPU_GICP_SERVER=0     every GICP cost evaluation is its own kernel launch (no resident server)
//   ICPGPU_GICP_DEVICE=0|1|auto   GICP's inner BFGS on the host (0), in the device solver gicp_solve_kernel (1); auto (the default)
//                            = the host loop for single alignments, the device solver for a batch's runs (fixed at creation since
//                            icpgpu.h 1.0; icpgpu_calibrate measures on request) -- same bits either way
//   ICPGPU_GICP_INNER=exact|quadratic   overrides icpgpu_params.gicp_inner (the ONE switch that moves a result: QUADRATIC stays
//                            within the stated tolerance of EXACT, not on its bits -- include/icpgpu.h: icpgpu_gicp_inner)
//   ICPGPU_MAILBOX=pairs|release   how results reach the host (default: self-test at context creation picks it)
//   ICPGPU_DEBUG=1           diagnostics on stderr
//   LOCAL_WORLD_SIZE         (torch.distributed.run) processes sharing this host's CPUs
// None of the others changes a result.
//
// DEVELOPMENT switches -- kernel variants, tuning constants, test modes that evaluate every pair, and a few that deliberately
// produce WRONG results to price a stage (ICPGPU_SKIP_UNCERT, *_NO_EXACT) -- exist only in the flavour compiled with
// -DICPGPU_DEV_SWITCHES (libicpgpu_dev.so, `make dev`; the tests of those modes load that flavour).  In the release
// library ICPGPU_DEV_ENV(name) is a constant null pointer: the switch, its getenv and its name are not in the binary.
// Round 6's: ICPGPU_COV_SELECT=0 (the streaming covariance kernel for every point), ICPGPU_COV_STATS=1 (what the selecting kernel
// did with each cloud), ICPGPU_VOXEL_PLANNED=0 (the voxel filter waits for the bounding box), ICPGPU_STAGE_DIRECT=0 (result clouds
// through device memory and the copy engine), ICPGPU_COV_GRID_UNCHECKED=0|2 (the covariance grid waits for its statistics / every
// check fails).
#pragma once
#include <cstdlib>

#if defined(ICPGPU_DEV_SWITCHES)
#define ICPGPU_DEV_ENV(name) (std::getenv(name))
#else
#define ICPGPU_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
