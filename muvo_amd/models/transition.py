"""Recurrent state-space model (muvo/models/transition.py:5-191) on HIP kernels.

Per time step: pre_gru Linear, GRUCell (two MFMA GEMMs + fused pointwise), prior / posterior MLPs, the
2*sigmoid(x/2)+0.1 / reparameterised-sample kernel.  `nn.LeakyReLU(True)` in the reference has
negative_slope = 1.0, i.e. it is the identity: no activation is applied here.
RNG is explicit: `noise` (b, s, 2, state_dim) [prior, posterior] and `use_prior` flags per step can be passed
in (parity tests); otherwise they are drawn like the reference does (torch.randn / one host coin per step)."""
import torch
import torch.nn as nn

from muvo_amd import nn as hnn
from muvo_amd import ops


class RepresentationModel(nn.Module):
    def __init__(self, in_channels, latent_dim):
        super().__init__()
        self.latent_dim = latent_dim
        self.min_std = 0.1
        self.module = nn.Sequential(hnn.Linear(in_channels, in_channels), hnn.Placeholder(),
                                    hnn.Linear(in_channels, 2 * latent_dim))

    def forward(self, x, eps):
        mls = self.module[2](self.module[0](x))
        return ops.rssm_sample(mls, eps, self.min_std)  # mu, sigma, sample


class RSSM(nn.Module):
    def __init__(self, embedding_dim, action_dim, hidden_state_dim, state_dim, action_latent_dim, receptive_field,
                 use_dropout=False, dropout_probability=0.0):
        super().__init__()
        self.embedding_dim, self.state_dim, self.action_dim = embedding_dim, state_dim, action_dim
        self.hidden_state_dim, self.action_latent_dim = hidden_state_dim, action_latent_dim
        self.receptive_field = receptive_field
        self.use_dropout, self.dropout_probability = use_dropout, dropout_probability
        self.pre_gru_net = nn.Sequential(hnn.Linear(state_dim, hidden_state_dim), hnn.Placeholder())
        self.recurrent_model = hnn.GRUCell(hidden_state_dim, hidden_state_dim)
        self.posterior_action_module = nn.Sequential(hnn.Linear(action_dim, action_latent_dim), hnn.Placeholder())
        self.posterior = RepresentationModel(hidden_state_dim + embedding_dim + action_latent_dim, state_dim)
        self.prior_action_module = nn.Sequential(hnn.Linear(action_dim, action_latent_dim), hnn.Placeholder())
        self.prior = RepresentationModel(hidden_state_dim + action_latent_dim, state_dim)
        self.active_inference = False

    def forward(self, input_embedding, action, use_sample=True, policy=None, noise=None, use_prior=None):
        b, s, _ = input_embedding.shape
        dev = input_embedding.device
        if use_sample and noise is None:
            noise = torch.randn(b, s, 2, self.state_dim, device=dev)
        if use_prior is None:
            use_prior = [bool(self.training and self.use_dropout and torch.rand(1).item() < self.dropout_probability
                              and t > 0) for t in range(s)]
        if use_sample and not self.active_inference and ops.rssm_fused_supported(
                b, s, self.hidden_state_dim, self.state_dim, self.embedding_dim, self.action_latent_dim, self.action_dim):
            # the whole time loop as one persistent kernel (csrc/rssm.hip); the hidden state is one tensor for both dicts
            h, p_mu, p_sigma, p_sample, q_mu, q_sigma, q_sample = ops.rssm_fused(input_embedding, action, noise, self, use_prior)
            return {'prior': {'hidden_state': h, 'sample': p_sample, 'mu': p_mu, 'sigma': p_sigma},
                    'posterior': {'hidden_state': h, 'sample': q_sample, 'mu': q_mu, 'sigma': q_sigma}}
        h_t = torch.zeros(b, self.hidden_state_dim, device=dev)
        sample_t = torch.zeros(b, self.state_dim, device=dev)
        zeros_a = torch.zeros(b, self.action_dim, device=dev)
        emb = ops.unstack_time(input_embedding)   # s contiguous (b, E) rows
        act = ops.unstack_time(action)
        keys = ('hidden_state', 'sample', 'mu', 'sigma')
        prior = {k: [] for k in keys}
        post = {k: [] for k in keys}
        for t in range(s):
            action_t = zeros_a if t == 0 else act[t - 1]
            # imagine step (transition.py:151-173)
            h_t = self.recurrent_model(self.pre_gru_net[0](sample_t), h_t)
            la_p = self.prior_action_module[0](action_t)
            p_mu, p_sigma, p_sample = self.prior(ops.cat_last([h_t, la_p]), noise[:, t, 0] if use_sample else None)
            # observe step (transition.py:130-149)
            la_q = self.posterior_action_module[0](action_t)
            q_mu, q_sigma, q_sample = self.posterior(ops.cat_last([h_t, emb[t], la_q]),
                                                     noise[:, t, 1] if use_sample else None)
            for d, vals in ((prior, (h_t, p_sample, p_mu, p_sigma)), (post, (h_t, q_sample, q_mu, q_sigma))):
                for k, v in zip(keys, vals):
                    d[k].append(v)
            sample_t = p_sample if use_prior[t] else q_sample
        return {'prior': {k: ops.stack_time(v) for k, v in prior.items()},
                'posterior': {k: ops.stack_time(v) for k, v in post.items()}}

    def observe_step(self, h_t, sample_t, action_t, embedding_t, use_sample=True, policy=None, eps=None, eps_prior=None):
        """transition.py:130-149: the imagine step, then the posterior from (h, embedding, action latent)."""
        prior = self.imagine_step(h_t, sample_t, action_t, use_sample, policy=policy, eps=eps_prior)
        la = self.posterior_action_module[0](action_t)
        if use_sample and eps is None:
            eps = torch.randn(h_t.shape[0], self.state_dim, device=h_t.device)
        mu, sigma, sample = self.posterior(ops.cat_last([prior['hidden_state'], embedding_t.contiguous(), la]), eps if use_sample else None)
        return {'prior': prior, 'posterior': {'hidden_state': prior['hidden_state'], 'sample': sample, 'mu': mu, 'sigma': sigma}}

    def imagine_step(self, h_t, sample_t, action_t, use_sample=True, policy=None, eps=None):
        """transition.py:151-173: one prior roll-out step (eps: the explicit N(0,1) draw, generated when None)."""
        if self.active_inference:
            action_t = policy(ops.cat_last([h_t, sample_t]))
        h_t = self.recurrent_model(self.pre_gru_net[0](sample_t), h_t)
        la = self.prior_action_module[0](action_t)
        if use_sample and eps is None:
            eps = torch.randn(h_t.shape[0], self.state_dim, device=h_t.device)
        mu, sigma, sample = self.prior(ops.cat_last([h_t, la]), eps if use_sample else None)
        return {'hidden_state': h_t, 'sample': sample, 'mu': mu, 'sigma': sigma}
