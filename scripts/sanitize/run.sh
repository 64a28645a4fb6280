#!/bin/bash
# AddressSanitizer + UBSan over the host-side native code (CPU build; GPU sanitizers are not available on the pool):
# the Motion-JPEG decoder on 60 000 mutated frames held in exact-size heap blocks, the CSV formatter on 20 000 random tables, the channel picker (r4) on 4 000 random sizes / channel counts / thread counts.
#   bash scripts/sanitize/run.sh        (needs Pillow to make the seed JPEGs; writes under /tmp/cbas_sanitize)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); W=/tmp/cbas_sanitize; mkdir -p $W; cd $W
python3 - <<'PY'
import numpy as np
from PIL import Image
rng = np.random.default_rng(0); i = 0
for ss in (0, 1, 2):
    for kw in ({}, {"restart_marker_blocks": 2}, {"optimize": True}):
        a = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
        Image.fromarray(a).save(f"/tmp/cbas_sanitize/b{i}.jpg", "JPEG", quality=int(rng.integers(20, 100)), subsampling=ss, **kw); i += 1
Image.fromarray(a[:, :, 0]).save(f"/tmp/cbas_sanitize/b{i}.jpg", "JPEG")
PY
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/cbas_amd/csrc"
$CXX $FLAGS -mavx2 $HERE/fuzz_mjpeg.cpp $ROOT/cbas_amd/csrc/host_mjpeg.cpp -o fuzz_mjpeg -lpthread
$CXX $FLAGS $HERE/fuzz_csv.cpp $ROOT/cbas_amd/csrc/host_text.cpp -o fuzz_csv -lpthread
$CXX $FLAGS $HERE/fuzz_pixels.cpp $ROOT/cbas_amd/csrc/host_pixels.cpp -o fuzz_pixels -lpthread
./fuzz_mjpeg b*.jpg
./fuzz_csv
./fuzz_pixels
