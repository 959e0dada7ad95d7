"""Module-constant hyper-parameters, mirroring mg/model/MusicTransformer/config.py:8-55.

Same names and default values as the reference (so ``train.py`` behaves the same with no flags);
the hard-coded ``/data2/qt/...`` paths of the reference are replaced by relative defaults."""
import torch

from .sequence import EventSeq

device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')

event_dim = EventSeq.dim()
pad_token = EventSeq.dim()
vocab_size = EventSeq.dim() + 1

save_path = './output/'
condition_file = None
length = 2000
threshold_len = 500

pickle_dir = './dataset/'

load_path = None
dropout = 0.2
debug = False

num_layers = 6
max_seq = 2048
embedding_dim = 256

l_r = 0.0001
batch_size = 6
window_size = 2048
stride_size = 10
accum_grad = 12
label_smooth = 0.1
epochs = 50000

model = {
    'vocab_size': vocab_size,
    'embedding_dim': embedding_dim,
    'max_seq': max_seq,
    'num_layer': num_layers,
    'dropout': dropout,
}

train = {
    'learning_rate': l_r,
    'batch_size': batch_size,
    'window_size': 2048,
    'stride_size': 10,
    'accum_grad': 25,
}
