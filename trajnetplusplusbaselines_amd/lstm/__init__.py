from .lstm import LSTM, LSTMPredictor, drop_distant
from .gridbased_pooling import GridBasedPooling
from .modules import Hidden2Normal, InputEmbedding
from .loss import PredictionLoss, L2Loss, CollisionLoss
