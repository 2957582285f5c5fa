"""The prediction file read on the device against the host reader at the
validation scale (development aid): three device reads and one host read of a
30 M-box file written by the native writer, the columns compared bit for bit.
    TAOAMD_INGEST_TIMING=1 python tools/ingest_prof.py        (on an MI355X box)
    rocprofv3 --kernel-trace --stats -- python tools/ingest_prof.py   (the js_* kernels)
"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from tao_amodal_amd.columns import DTColumns
from tao_amodal_amd import columns
from tao_amodal_amd.synth import synth
path = '/tmp/_pred_prof.json'
if not os.path.exists(path):
    gt, dt = synth(seed=20240807, V=int(os.environ.get('V', '2000')), F=300, C=1203, dets_per_frame=50)
    dt.write_json(path)
    del gt, dt
print('file MB', os.path.getsize(path) / 1e6)
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(columns.__file__)), "libtao_amodal_ingest.so"))
torch.zeros(1, device='cuda')
for rep in range(3):
    t = time.perf_counter()
    d = DTColumns._from_file_device(path, lib)
    print('device ingest %.3f s, n = %d' % (time.perf_counter() - t, len(d)))
os.environ["TAOAMD_DEVICE_INGEST"] = "0"
t = time.perf_counter()
h = DTColumns.from_file_native(path)
print('host ingest %.3f s' % (time.perf_counter() - t))
for f in DTColumns.FIELDS:
    assert (np.asarray(getattr(d, f)).view(np.uint64) == np.asarray(getattr(h, f)).view(np.uint64)).all(), f
print('identical')
