from .lstm import LSTM, LSTMPredictor, drop_distant
from .gridbased_pooling import GridBasedPooling
from .non_gridbased_pooling import (NearestNeighborMLP, HiddenStateMLPPooling, AttentionMLPPooling, NearestNeighborLSTM,
                                    TrajectronPooling)
from .modules import Hidden2Normal, InputEmbedding
from .loss import PredictionLoss, L2Loss, CollisionLoss, bce_loss, gan_g_loss, gan_d_loss, variety_loss
