from .sgan import SGAN, LSTMGenerator, LSTMDiscriminator, SGANPredictor, get_noise, make_mlp
