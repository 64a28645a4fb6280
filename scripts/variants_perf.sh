#!/bin/bash
# Throughput of the other encoder configurations + the host-streamed path
python scripts/quick_perf.py vits16 64 20
python scripts/quick_perf.py vitb16 64 20
python scripts/quick_perf.py vitb16 64 20 256
python scripts/quick_perf.py vitl16 32 5 518
python scripts/quick_perf.py dinov2regb14 64 20 224
python scripts/quick_perf.py dinov2regb14 64 20 252
python scripts/quick_perf.py vitb16 64 20 224 1
python scripts/host_stream_perf.py 4096
