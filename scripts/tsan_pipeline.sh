#!/bin/bash
# ThreadSanitizer run of the host pipeline (CPU only: the orchestration over the oracle backend, frontend's step on its worker thread).
# Builds everything with -fsanitize=thread into /tmp, renders an 80-frame half-resolution stream with a moving object (RD path on)
# and replays it with threading = 2.  Exit code 0 and no "WARNING: ThreadSanitizer" on stderr = no data race on the paths taken.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/rdvio_tsan
mkdir -p $O
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from rd_vio_amd import synth
W, H = 376, 240
K = synth.EUROC_K.copy(); K[:2] *= 0.5
frames, ts, imu, gt = synth.make_stream(80, W, H, K, mover=True)
with open("$O/stream.bin", "wb") as f:
    np.array([len(ts), W, H, len(imu), 2, 1], dtype=np.int32).tofile(f)
    for a in (K, synth.EUROC_EXTR, synth.EUROC_NOISE, ts, imu, gt):
        np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    np.ascontiguousarray(frames, dtype=np.uint8).tofile(f)
print("stream written")
PY
SAN="-fsanitize=thread -g -O1 -fno-omit-frame-pointer"
for c in ro_math ro_image ro_estimation ro_solver; do gcc -std=c99 $SAN -ffp-contract=off -c $R/oracle/$c.c -o $O/$c.o; done
gcc -std=c99 $SAN -c $R/oracle/backend/oracle_backend.c -o $O/oracle_backend.o -I$R/include
SRC=$(python3 -c "import sys; sys.path.insert(0, '$R'); from rd_vio_amd import build; import os; print(' '.join(os.path.join(build.PIPE_DIR, s) for s in build.PIPE_SOURCES))")
g++ -std=c++17 $SAN -ffp-contract=off -pthread -o $O/tsan_driver $R/tests/cpp/tsan_pipeline_driver.cpp $SRC $O/*.o -L$R/rd_vio_amd -lrdvio_hip -Wl,-rpath,$R/rd_vio_amd -Wl,-rpath,/opt/rocm/lib -lm
TSAN_OPTIONS="halt_on_error=0 exitcode=66" $O/tsan_driver $O/stream.bin
