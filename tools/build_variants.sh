#!/bin/bash
# Builds experimental variants of the GC-ADPCM kernels (different -D switches) as full libraries under
# tools/variants/ (git-ignored, shipped to the GPU box by gpurun).  usage: build_variants.sh name:"-DX -DY" ...
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fwrapv -fno-fast-math -w"
VARIED="${VARIED:-gc_encode_kernel gcadpcm_kernels hca_encode_kernel hca_decode_kernels}"
mkdir -p tools/variants/obj
for f in vgaudio_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  case " $VARIED " in *" $b "*) continue;; esac
  if [ ! -f tools/variants/obj/$b.o ] || [ $f -nt tools/variants/obj/$b.o ]; then
    /opt/rocm/bin/hipcc $FLAGS $(python3 vgaudio_amd/build.py --file-flags $b.hip) -c $f -o tools/variants/obj/$b.o &
  fi
done
wait
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  for v in $VARIED; do
    # (the product's per-file options, vgaudio_amd/build.py FILE_FLAGS, unless the variant says NOFILEFLAGS=1)
    /opt/rocm/bin/hipcc $FLAGS $([ -n "$NOFILEFLAGS" ] || python3 vgaudio_amd/build.py --file-flags $v.hip) $defs -c vgaudio_amd/csrc/$v.hip -o tools/variants/obj/var_${name}_$v.o &
  done
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  others=$(ls tools/variants/obj/*.o | grep -v "/var_")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others tools/variants/obj/var_${name}_*.o -o tools/variants/libvga_$name.so
  echo built tools/variants/libvga_$name.so
done
