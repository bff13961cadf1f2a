# Builds the diagnostic library (image.hip with -DCTVAE_PHASE_TIMING) on the GPU box and prints img_fwd_kernel's phase timeline.
set -e
cd $GRAFT_REPO_ROOT
O=ct-vae_amd/csrc/_obj
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCTVAE_PHASE_TIMING -c ct-vae_amd/csrc/image.hip -o /tmp/image_t.o
objs=$(ls $O/*.o | grep -v "/image.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libctvae_timing.so $objs /tmp/image_t.o
CTVAE_TIMING_LIB=/tmp/libctvae_timing.so python tools/img_fwd_phase_probe.py
