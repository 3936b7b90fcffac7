import sys, torch
sys.path.insert(0, '.')
from muvo_amd import ops
from muvo_amd.models.transition import RSSM
dev = torch.device('cuda:0')
with torch.device(dev):
    m = RSSM(embedding_dim=32, action_dim=2, hidden_state_dim=64, state_dim=32, action_latent_dim=8, receptive_field=3,
             use_dropout=True, dropout_probability=0.15).train()
for p in m.parameters():
    p.grad = torch.zeros_like(p)
for it in range(12):
    emb = torch.randn(2, 3, 32, device=dev, requires_grad=True)
    act = torch.rand(2, 3, 2, device=dev)
    eps = torch.randn(2, 3, 2, 32, device=dev)
    try:
        out = m(emb, act, noise=eps, use_prior=[False, True, False])
        (out['posterior']['sample'].sum() + out['prior']['mu'].sum()).backward()
        print('iter', it, 'ok', flush=True)
    except RuntimeError as e:
        print('iter', it, 'FAILED', str(e)[:200], flush=True)
torch.cuda.synchronize()
