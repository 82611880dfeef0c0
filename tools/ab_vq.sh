L=$PWD/interactive-spectrogram-inpainting_amd
LIBS="${LIBS:-lib lib_exp}"
export ISI_HIP_LIBRARY=$L/lib_exp/libisi_hip.so
python -m pytest tests/test_hip_parity.py -q -x -k "vq or quant or codebook or nearest or golden or full_size or certified or far" 2>&1 | tail -2
for i in 1 2; do
for lib in $LIBS; do
ISI_HIP_LIBRARY=$L/$lib/libisi_hip.so python bench.py --no-cpu-baseline --no-prior --no-train --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], [ (k['kernel'][:12], k['ms_per_step']) for k in d['kernels']])"
done; done
