"""Drop-in for mg/model/MusicTransformer/generate.py:18-123: load a checkpoint, print a 2-sample test
loss/accuracy, sample ``--max-length`` events from a prior and write them out.
MIDI-like and REMI samples are written as .mid files (pretty_midi if installed, else the built-in SMF writer,
smf.py; REMI / MuMIDI through their write_midi on the same writer)."""
from __future__ import annotations

import optparse
import os

import numpy as np
import torch

from . import config, utils
from .criterion import SmoothCrossEntropyLoss
from .data import Data
from .metrics import CategoricalAccuracy, LogitsBucketting, MetricsSet
from .network import MusicTransformer
from .train import vocab_of


def get_options(argv=None):
    parser = optparse.OptionParser()
    parser.add_option('-b', '--batch-size', dest='batch_size', type='int', default=8)
    parser.add_option('-s', '--load_path', dest='load_path', type='string', default=None)
    parser.add_option('-o', '--output-dir', dest='output_dir', type='string', default='./output/generate/')
    parser.add_option('-d', '--dataset', dest='data_path', type='string', default=config.pickle_dir)
    parser.add_option('-l', '--max-length', dest='max_len', type='int', default=config.length)
    parser.add_option('-T', '--temperature', dest='temperature', type='float', default=1.0)
    parser.add_option('--top-k', dest='top_k', type='int', default=0)
    parser.add_option('--top-p', dest='top_p', type='float', default=1.0)
    parser.add_option('--num-layers', dest='num_layers', type='int', default=config.num_layers)
    parser.add_option('--d-model', dest='d_model', type='int', default=config.embedding_dim)
    parser.add_option('--repr', dest='repr', type='string', default='midi_like')
    parser.add_option('--grammar', dest='grammar', action='store_true', default=False,
                      help='constrain sampling to the REMI / MuMIDI event grammar (KV-cache decode, mask inside the sampler)')
    parser.add_option('--reference-mask', dest='reference_mask', action='store_true', default=False,
                      help="sample exactly as the reference's generate() does: Decoder(window, mask=None), i.e. no look-ahead "
                           'mask at sampling time (network.py:60); default: the training-time causal semantics')
    parser.add_option('-M', '--max_seq', dest='max_seq', type='int', default=config.max_seq)
    parser.add_option('-c', '--condition-file', dest='condition_file', type='string', default=getattr(config, 'condition_file', None),
                      help='MIDI file to continue (the reference reads config.condition_file, generate.py:101-105): its '
                           'first 500 MIDI-like events become the prior of every sample')
    return parser.parse_args(argv)[0]


def main(argv=None):
    o = get_options(argv)
    device = torch.device('cuda:0')
    vocab = vocab_of(o.repr)
    mt = MusicTransformer(embedding_dim=o.d_model, vocab_size=vocab, num_layer=o.num_layers, max_seq=o.max_seq,
                          dropout=0)
    if o.load_path:
        mt.load_state_dict(torch.load(o.load_path, map_location='cpu', weights_only=False)['net'])
    mt.to(device).eval()
    if o.data_path and os.path.isdir(o.data_path):
        ds = Data(o.data_path, o.max_seq)
        if len(ds.file_dict['test']) >= 2:
            ms = MetricsSet({'accuracy': CategoricalAccuracy(), 'loss': SmoothCrossEntropyLoss(config.label_smooth, vocab, vocab - 1),
                             'bucket': LogitsBucketting(vocab)})
            x, y = ds.slide_seq2seq_batch(2, o.max_seq, 'test')
            with torch.no_grad():
                pred, _ = mt(torch.from_numpy(x).to(device, dtype=torch.int))
                m = ms(pred, torch.from_numpy(y).to(device, dtype=torch.int))
            print('Test >>>> Loss: {:6.6}, Accuracy: {}'.format(m['loss'], m['accuracy']))
    mt.test()
    prior = torch.tensor([[24, 28, 31]] * o.batch_size, dtype=torch.long, device=device)
    if o.condition_file is not None:
        # generate.py:101-105: MIDI -> notes -> MIDI-like events -> the first 500 indices, repeated for the batch
        if o.repr != 'midi_like':
            raise SystemExit('--condition-file continues a MIDI-like (EventSeq) prompt: use --repr midi_like')
        from .sequence import EventSeq, NoteSeq
        ids = EventSeq.from_note_seq(NoteSeq.from_midi_file(o.condition_file)).to_array()[:500]
        if len(ids) == 0:
            raise SystemExit(f'{o.condition_file}: no notes in the MIDI-like pitch range')
        prior = torch.from_numpy(np.array([ids] * o.batch_size, dtype=np.int64)).to(device)
        print('Prompt: {} events from {}'.format(len(ids), o.condition_file))
    if o.grammar:
        if o.repr == 'remi':
            from .REMI import REMI_EventSeq as Codec
        elif o.repr == 'mumidi':
            from .MuMIDI import MuMIDI_EventSeq as Codec
        else:
            raise SystemExit('--grammar is defined for --repr remi and --repr mumidi')
        bar = Codec.feat_ranges()['bar'][0]
        prior = torch.full((o.batch_size, 1), bar, dtype=torch.long, device=device)
        res = mt.generate_cached(prior, o.max_len, temperature=o.temperature, top_k=o.top_k, top_p=o.top_p,
                                 grammar=Codec.next_token_table()).cpu().numpy()
    else:
        res = mt.generate(prior, o.max_len, temperature=o.temperature, top_k=o.top_k, top_p=o.top_p,
                          reference_mask=o.reference_mask).cpu().numpy()
    os.makedirs(o.output_dir, exist_ok=True)
    for i, seq in enumerate(res):
        name = os.path.join(o.output_dir, f'gen-{i:03d}')
        if o.repr == 'midi_like':
            n = utils.event_indeces_to_midi_file(seq, name + '.mid')
            print('===> {} ({} notes)'.format(name + '.mid', n))
        elif o.repr == 'remi':
            from .REMI import REMI_EventSeq
            ids = [int(v) for v in seq if int(v) < REMI_EventSeq.dim()]            # drop pad ids
            notes, _, _ = REMI_EventSeq.write_midi(REMI_EventSeq.to_event(ids), name + '.mid')
            print('===> {} ({} notes)'.format(name + '.mid', len(notes)))
        elif o.repr == 'mumidi':
            from .MuMIDI import MuMIDI_EventSeq
            ids = [int(v) for v in seq if int(v) < MuMIDI_EventSeq.dim()]          # drop pad ids
            notes, _, _ = MuMIDI_EventSeq.write_midi(MuMIDI_EventSeq.from_array(ids), name + '.mid')
            print('===> {} ({} notes)'.format(name + '.mid', sum(len(v) for v in notes.values())))
        else:
            np.save(name + '.npy', seq.astype(np.uint16))
            print('===> {} (event indices)'.format(name + '.npy'))


if __name__ == '__main__':
    main()
