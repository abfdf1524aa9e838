mkdir -p gpurun_out/r02n
cp ddsp_amd/lib/libddsp_amd.so /tmp/base.so
for v in base ONE_MFMA NO_DPP; do
  if [ $v = base ]; then cp /tmp/base.so ddsp_amd/lib/libddsp_amd.so; else cp ddsp_amd/lib/x_$v.so ddsp_amd/lib/libddsp_amd.so; fi
  python - <<'PY'
from ddsp_amd import build
open(build.STAMP,'w').write(build.source_digest()+'\n')
PY
  for k in 0 1; do echo "== $v skip $k"; DDSP_MF_DBG_SKIP=$k DDSP_MF_DBG_WAVE=13 timeout 100 python tools/exp_noise_fir.py 128 2>&1 | grep -E "tick  [45]|kernel_us" | cut -c1-140; done
done
cp /tmp/base.so ddsp_amd/lib/libddsp_amd.so
