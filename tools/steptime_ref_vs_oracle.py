import sys, time, torch
sys.path.insert(0, "/root/repo")
from oracle import ref_import, lm as O
import ssr_speech_amd
from ssr_speech_amd import weights as W
torch.set_num_threads(8)
ssr = ref_import.import_lm()
args = W.lm_args_830m()
sd = W.lm_state_dict(args, seed=0)
m = ssr.SSR_Speech(args).eval()
m.load_state_dict(sd, strict=False)
g = torch.Generator().manual_seed(2024)
L, N, steps = 130, 160, 25
x = torch.randint(0, 100, (1, L), generator=g); y = torch.randint(0, 2048, (1, N, 4), generator=g)
mi = torch.LongTensor([[[N, N]]])
kw = dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, kvcache=1, cfg_coef=1.5, cfg_stride=5, aug_text=True)
marks = []
orig = ssr.topk_sampling
class Stop(Exception): pass
def spy(*a, **k):
    marks.append(time.perf_counter())
    if len(marks) > steps: raise Stop()
    return orig(*a, **k)
ssr.topk_sampling = spy
torch.manual_seed(1)
try:
    with torch.no_grad(): m.inference(x, torch.LongTensor([L]), x, torch.LongTensor([L]), y, y, mi, **kw)
except Stop: pass
ssr.topk_sampling = orig
import numpy as np
d = np.diff(np.asarray(marks))[2:]
print("reference ms/step", 1000 * d.mean())
sdo = O.reference_params(sd)
marks2 = []
class Clock(dict):
    def setdefault(self, k, dflt=None):
        if k == "samples": marks2.append(time.perf_counter())
        return dict.setdefault(self, k, dflt)
torch.manual_seed(1)
unc = torch.randint(0, 101, (1, L))
O.inference(sdo, args, x, y, mi, uncond_x=unc, max_steps=steps, trace=Clock(), **kw)
d2 = np.diff(np.asarray(marks2))[2:]
print("oracle ms/step", 1000 * d2.mean())
