"""Drop-in for mg/model/Event_MelodyRNN/train.py: same optparse flags (train.py:22-100), same loop for the
``segment`` mode (train.py:327-362, the reference's configured mode) and for ``window`` with teacher forcing 1.0
(train.py:217-262, where ``generate(..., output_type='logit')`` equals ``Train``): random init vector -> ``Train`` ->
cross-entropy -> ``clip_grad_norm_`` -> Adam, one ``state_dict`` checkpoint per epoch
(``{mode}_512_3_1_epoch_{n}.pth``, train.py:188-195).  ``sequence`` mode (train.py:263-287): whole variable-length
sequences, ``SeqBatchify`` collate (sorted, zero-padded ``X [B,Tmax]``, concatenated labels ``X[i,1:len_i]``),
``Train(init, X, lengths)`` and the loss over ``flatten_padded_sequences`` -- the computation the reference's loop is
written for (its own ``SeqForward`` mixes the batch and time axes and cannot run).  Not built: teacher forcing < 1
(sampling inside the training graph)."""
from __future__ import annotations

import optparse
import os
import time

import numpy as np
import torch
from torch import nn, optim

from . import utils
from .data import Event_Dataset, MyDataset, SeqBatchify, flatten_padded_sequences
from .melody_rnn import Event_Melody_RNN
from .sequence import EventSeq

# Event_MelodyRNN/config.py
TRAIN_MODE = "segment"
LIMLEN = 1200
MODEL = {'init_dim': 32, 'event_dim': EventSeq.dim(), 'hidden_dim': 512, 'rnn_layers': 3, 'dropout': 0.3}
TRAIN = {'learning_rate': 0.001, 'batch_size': 100, 'window_size': 200, 'stride_size': 10, 'use_transposition': False,
         'teacher_forcing_ratio': 1.0, 'clip_norm': 1.0}


def get_options(argv=None):
    parser = optparse.OptionParser()
    parser.add_option('-s', '--save_path', dest='save_path', type='string', default='./save_model/')
    parser.add_option('-d', '--dataset', dest='data_path', type='string', default='./data/')
    parser.add_option('-e', '--epochs', dest='epochs', type='int', default=200)
    parser.add_option('-i', '--saving-interval', dest='saving_interval', type='float', default=60.)
    parser.add_option('-b', '--batch-size', dest='batch_size', type='int', default=TRAIN['batch_size'])
    parser.add_option('-l', '--learning-rate', dest='learning_rate', type='float', default=TRAIN['learning_rate'])
    parser.add_option('-w', '--window-size', dest='window_size', type='int', default=TRAIN['window_size'])
    parser.add_option('-S', '--stride-size', dest='stride_size', type='int', default=TRAIN['stride_size'])
    parser.add_option('-T', '--teacher-forcing-ratio', dest='teacher_forcing_ratio', type='float',
                      default=TRAIN['teacher_forcing_ratio'])
    parser.add_option('-n', '--clip_norm', dest='clip_norm', type='float', default=TRAIN['clip_norm'])
    parser.add_option('-t', '--use-transposition', dest='use_transposition', action='store_true',
                      default=TRAIN['use_transposition'])
    parser.add_option('-p', '--model-params', dest='model_params', type='string', default='')
    parser.add_option('-r', '--reset-optimizer', dest='reset_optimizer', action='store_true', default=False)
    parser.add_option('-L', '--enable-logging', dest='enable_logging', action='store_true', default=False)
    parser.add_option('-q', '--limit-length', dest='limlen', type='int', default=LIMLEN)
    parser.add_option('--mode', dest='mode', type='string', default=TRAIN_MODE, help="segment | window | sequence (config.train_mode)")
    return parser.parse_args(argv)[0]


def main(argv=None):
    o = get_options(argv)
    if o.mode not in ('segment', 'window', 'sequence'):
        raise ValueError("--mode must be segment, window or sequence")
    if o.mode == 'window' and o.teacher_forcing_ratio != 1.0:
        raise NotImplementedError("teacher forcing < 1 samples inside the training graph: not built")
    model_config = dict(MODEL)
    for k, v in utils.params2dict(o.model_params).items():
        model_config[k] = type(MODEL.get(k, v))(v)
    device = torch.device('cuda:0')
    event_dim = model_config['event_dim']
    print('-' * 70)
    print('Save path:', o.save_path)
    print('Dataset path:', o.data_path)
    print('Hyperparameters:', utils.dict2params(model_config))
    print('Learning rate:', o.learning_rate)
    print('Batch size:', o.batch_size)
    print('-' * 70)
    model = Event_Melody_RNN(**model_config).to(device)
    optimizer = optim.Adam(model.parameters(), lr=o.learning_rate)
    dataset = Event_Dataset(o.data_path, o.limlen, verbose=True)
    assert len(dataset.samples) > 0
    print(dataset)
    window, stride = o.window_size, o.stride_size
    if o.mode == 'segment':                                   # train.py:328-333
        window = int(np.min(dataset.seqlens))
        stride = max(1, window // 3)
        print(f'Window Size = {window}')
        print(f'Stride = {stride}')
    if o.mode == 'sequence':                                  # train.py:263-272: whole sequences, packed by length
        loader = torch.utils.data.DataLoader(MyDataset(dataset.samples), o.batch_size, collate_fn=SeqBatchify, shuffle=True,
                                             drop_last=True, num_workers=0)
        print(f'Iteration={len(dataset.samples) // o.batch_size}')
    else:
        windows = dataset.batches(o.batch_size, window, stride)
        print(f'Iteration={len(windows) // o.batch_size}')
        loader = torch.utils.data.DataLoader(MyDataset(windows), o.batch_size, collate_fn=dataset.SegBatchify, shuffle=True,
                                             drop_last=True, num_workers=0)
    loss_function = nn.CrossEntropyLoss()
    os.makedirs(o.save_path, exist_ok=True)

    def save_model(epoch):
        path = os.path.join(o.save_path, f'{o.mode}_512_3_1_epoch_{epoch}.pth')
        print('Saving to', path)
        torch.save(model.state_dict(), path)
        print('Done saving')

    last = time.time()
    model.train()
    for epoch in range(o.epochs):
        try:
            l_sum, n = 0.0, 0
            for iteration, batch in enumerate(loader):
                init = torch.randn(o.batch_size, model.init_dim, device=device)
                if o.mode == 'sequence':
                    X, label, lengths = batch                                                         # [B,Tmax], [sum(len-1)], [B]
                    X = torch.from_numpy(np.ascontiguousarray(X).astype(np.int64)).to(device)
                    label = torch.from_numpy(np.ascontiguousarray(label).astype(np.int64)).to(device)
                    outputs = model.Train(init, events=X, lengths=lengths)                            # [B, Tmax+1, V]
                    # step t+1 has consumed X[i, 0..t] and predicts X[i, t+1]: rows 1..len-1 of each sample carry a label
                    loss = loss_function(flatten_padded_sequences(outputs[:, 1:], lengths), label)
                else:
                    events = torch.from_numpy(np.ascontiguousarray(batch).astype(np.int64)).to(device)     # [T, B]
                    outputs = model.Train(init, events=events[:-1])
                    loss = loss_function(outputs.view(-1, event_dim), events.view(-1))
                model.zero_grad()
                loss.backward()
                l_sum += loss.item()
                n += 1
                nn.utils.clip_grad_norm_(model.parameters(), max_norm=o.clip_norm if o.mode == 'window' else 1.0)
                optimizer.step()
                if (iteration + 1) % 50 == 0:
                    print(f'epoch {epoch}, iter {iteration}, loss: {loss.item()}')
            print(f'epoch {epoch}, ave-loss: {l_sum / max(n, 1)}, epoch time: {time.time() - last}')
            last = time.time()
            save_model(epoch)
        except KeyboardInterrupt:
            save_model(epoch)
            break
    return model


if __name__ == '__main__':
    main()
