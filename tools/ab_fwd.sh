L=$PWD/interactive-spectrogram-inpainting_amd
LIBS="${LIBS:-lib lib_exp lib_exp2}"
for i in 1 2; do
for lib in $LIBS; do
ISI_HIP_LIBRARY=$L/$lib/libisi_hip.so python bench.py --no-cpu-baseline --no-prior --no-train --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], [ (k['kernel'][:12], k['ms_per_step']) for k in d['kernels']])"
done; done
