"""KLDLoss of the reference's VAE (vae/loss.py:6-52): a handful of elementwise operations on the primaries' [B, 2 latent_dim]
latent statistics -- tensor expressions on the device, nothing a kernel would improve."""
import torch


class KLDLoss(torch.nn.Module):
    def forward(self, inputs, batch_split, targets=None):
        """inputs [M, 2 latent_dim] = (z_mu | z_log_var); targets the same layout or None (standard normal).  Primaries only
        (rows batch_split[:-1]), mean over scenes -- reference vae/loss.py:13-52."""
        prim = torch.as_tensor(batch_split, dtype=torch.int64)[:-1].to(inputs.device)
        inputs = inputs[prim]
        z_mu, z_log_var = torch.split(inputs, inputs.size(1) // 2, dim=1)
        if targets is None:
            latent_loss = -0.5 * torch.sum(1.0 + z_log_var - torch.square(z_mu) - torch.exp(z_log_var), dim=1)
        else:
            targets = targets[prim]
            z_mu_t, z_log_var_t = torch.split(targets, targets.size(1) // 2, dim=1)
            z_var, z_var_t = torch.exp(z_log_var), torch.exp(z_log_var_t)
            latent_loss = 0.5 * (((1 / z_var_t) * z_var).sum(dim=1) + ((z_mu_t - z_mu) ** 2 * (1 / z_var_t)).sum(dim=1))
        return torch.mean(latent_loss)
