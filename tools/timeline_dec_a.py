import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
tl = torch.zeros(64, dtype=torch.int64, device='cuda')
s = torch.randn(19200, 10, device='cuda') * 0.3
m.model_down.decoder(s); torch.cuda.synchronize()
m.set_option('tl_buf', tl.data_ptr())
m.model_down.decoder(s); torch.cuda.synchronize()
t = tl.cpu().numpy()
n = int((t != 0).sum())
d = t[1:n] - t[:n-1]
print('stamps', n)
names = ['stage+bar', 'pf issue+CT1', 'bar', 'writeback+bar', 'par3..', ]
print([int(x) for x in d])
