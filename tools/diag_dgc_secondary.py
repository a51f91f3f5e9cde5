import sys, time, torch
sys.path[:0]=['/root/repo/deeprob-kit_amd','/root/repo']
from deeprob.spn.models import DgcSpn
from oracle import dgcspn_oracle as dorc
torch.manual_seed(5)
m = DgcSpn((1,28,28), n_batch=16, sum_channels=32, depthwise=True, n_pooling=2).eval()
sd = {k:v.detach().clone() for k,v in m.state_dict().items()}
print(sum(p.numel() for p in m.parameters()), [type(l).__name__ for l in m.layers][:12])
m.cuda()
B=8192
x=torch.randn(B,1,28,28,device='cuda')
with torch.no_grad():
    for _ in range(3): y=m(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): y=m(x)
    torch.cuda.synchronize(); print('ms', (time.perf_counter()-t0)/10*1e3)
plan = dorc.schedule((1,28,28),16,32,True,2)
want = dorc.dgcspn_forward(sd, x[:64].cpu(), plan)
print('rel', float(((y[:64].cpu()-want).abs()/want.abs().clamp_min(1)).max()))
