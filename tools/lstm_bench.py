import sys, os, time
sys.path.insert(0, "video-text-detection-system_amd"); sys.path.insert(0, ".")
import torch
from vtd_amd import nets
from vtd_amd.engine import RecognizerEngine
sd = nets.seeded_state_dict(lambda: nets.CRNN(97), seed=11)
eng = RecognizerEngine(97, sd, max_crops=512)
x = torch.rand(int(os.environ.get("REC_CROPS", "272")), 3, 32, 128)
for _ in range(3): eng.forward_logits(x)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(20): eng.forward_logits(x)
torch.cuda.synchronize()
print("dbg", os.environ.get("VTD_LSTM_DBG"), "forward ms", (time.perf_counter()-t)/20*1e3)
