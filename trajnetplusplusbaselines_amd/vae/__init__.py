"""VAE forecaster of the reference (trajnetbaselines/vae) on the MI355X sequence driver."""
from .vae import VAE, VAEPredictor, VAEEncoder, VAEDecoder, drop_distant  # noqa: F401
from .loss import KLDLoss  # noqa: F401
