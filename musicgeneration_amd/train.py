"""Drop-in for mg/model/MusicTransformer/train.py: same optparse flags (-s -d -e -i -b -l -w -S -g -m -M),
same prints, same checkpoint dict {'net','optimizer','epoch'} and file name, same loop semantics
(train.py:252-329: per-epoch batches, loss/accum_grad, Noam schedule step every accum_grad
micro-batches, 2-sample eval per epoch, save every 50 epochs and on Ctrl-C) -- running on the MI355X
kernels.  Extras that do not change the defaults: --num-layers --d-model --repr --dropout --dp, and the
data-parallel co-residency knobs --rccl-cus --nccl-channels --buckets.

The reference parses -b/-e/-l/-w/-S/-i/-g but its loop reads config.* instead; here the
flags are honoured, with the reference's config constants as their defaults.  Unlike the reference
this is a function (``main(argv)``), not code executed at import.

Multi-GPU (``--multi_gpu True`` or --dp): launch with torchrun; one process per GPU, RCCL all-reduce
of the flat gradient buffer (musicgeneration_amd/dp.py)."""
from __future__ import annotations

import optparse
import os
import random
import time

import torch

from . import config, utils
from .criterion import CustomSchedule, SmoothCrossEntropyLoss
from .data import Data
from .metrics import CategoricalAccuracy, LogitsBucketting, MetricsSet
from .network import MusicTransformer
from .optim import FusedAdam


def get_options(argv=None):
    parser = optparse.OptionParser()
    parser.add_option('-s', '--save_path', dest='save_path', type='string', default=config.save_path)
    parser.add_option('-d', '--dataset', dest='data_path', type='string', default=config.pickle_dir)
    parser.add_option('-e', '--epochs', dest='epochs', type='int', default=config.epochs)
    parser.add_option('-i', '--saving-interval', dest='saving_interval', type='int', default=50)
    parser.add_option('-b', '--batch-size', dest='batch_size', type='int', default=config.batch_size)
    parser.add_option('-l', '--learning-rate', dest='learning_rate', type='float', default=config.train['learning_rate'])
    parser.add_option('-w', '--window-size', dest='window_size', type='int', default=config.train['window_size'])
    parser.add_option('-S', '--stride-size', dest='stride_size', type='int', default=config.train['stride_size'])
    parser.add_option('-g', '--multi_gpu', dest='multi_gpu', type='string', default='False')
    parser.add_option('-m', '--load_path', dest='load_path', type='string', default=None)
    parser.add_option('-M', '--max_seq', dest='max_seq', type='int', default=config.max_seq)
    # extras (absent from the reference; defaults reproduce it)
    parser.add_option('--num-layers', dest='num_layers', type='int', default=config.num_layers)
    parser.add_option('--d-model', dest='d_model', type='int', default=config.embedding_dim)
    parser.add_option('--repr', dest='repr', type='string', default='midi_like', help='midi_like | remi | mumidi')
    parser.add_option('--dropout', dest='dropout', type='float', default=config.dropout)
    parser.add_option('--accum-grad', dest='accum_grad', type='int', default=config.accum_grad)
    parser.add_option('--field', dest='field', type='string', default=None, help="MuMIDI .data dict field")
    parser.add_option('--max-batches', dest='max_batches', type='int', default=0, help='stop after N micro-batches')
    # data-parallel co-residency knobs (the same three as bench.py): CUs kept free of the compute stream for
    # RCCL's kernels, the number of RCCL channels, per-layer gradient buckets merged into K groups
    parser.add_option('--rccl-cus', dest='rccl_cus', type='int', default=0, help='multiple of 8; 0 = the compute stream uses the whole chip')
    parser.add_option('--nccl-channels', dest='nccl_channels', type='int', default=0, help="0 = RCCL's default")
    parser.add_option('--buckets', dest='buckets', type='int', default=0, help='0 = one all-reduce per layer')
    return parser.parse_args(argv)[0]


def vocab_of(repr_name):
    if repr_name == 'midi_like':
        from .sequence import EventSeq
        return EventSeq.dim() + 1
    if repr_name == 'remi':
        from .REMI import REMI_EventSeq
        return REMI_EventSeq.dim() + 1
    if repr_name == 'mumidi':
        from .MuMIDI import MuMIDI_EventSeq
        return MuMIDI_EventSeq.dim() + 1
    raise ValueError(repr_name)


def main(argv=None):
    options = get_options(argv)
    prev = torch.cuda.current_stream() if torch.cuda.is_available() else None
    try:
        return _run(options)
    finally:
        if options.rccl_cus > 0 and prev is not None:      # --rccl-cus made a masked stream the current one: hand the caller its own back
            from . import ops
            torch.cuda.synchronize()
            torch.cuda.set_stream(prev)
            ops.configure_streams(0, 0)


def _run(options):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi_gpu = options.multi_gpu == 'True' or world > 1
    if not torch.cuda.is_available():
        raise SystemExit("train.py runs on the MI355X kernels: no HIP device visible")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if options.nccl_channels > 0:                      # before the communicator exists: RCCL reads them at init
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(options.nccl_channels)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if options.rccl_cus > 0:
        # every kernel of this process from here on runs on a stream restricted to the other CUs (ops.configure_streams)
        from . import ops
        ops.configure_streams(0, options.rccl_cus, device=device)
        torch.cuda.set_stream(ops.main_stream(device))

    vocab = vocab_of(options.repr)
    pad = vocab - 1
    model_config = {'vocab_size': vocab, 'embedding_dim': options.d_model, 'max_seq': options.max_seq,
                    'num_layer': options.num_layers, 'dropout': options.dropout}
    save_path, load_path = options.save_path, options.load_path
    log = print if rank == 0 else (lambda *a, **k: None)
    log('-' * 70)
    log('Save path:', save_path)
    log('Dataset path:', options.data_path)
    log('Saving interval:', options.saving_interval)
    log('-' * 70)
    log('Hyperparameters:', utils.dict2params(model_config))
    log('Learning rate:', options.learning_rate)
    log('Batch size:', options.batch_size)
    log('Window size:', options.window_size)
    log('Stride size:', options.stride_size)
    log('Device:', device)
    log('-' * 70)

    metric_set = MetricsSet({
        'accuracy': CategoricalAccuracy(),
        'loss': SmoothCrossEntropyLoss(config.label_smooth, vocab, pad),
        'bucket': LogitsBucketting(vocab)})

    log(options.data_path)
    # under data parallelism a rank must never skip a micro-batch alone (the others would wait in the all-reduce):
    # drop the files whose random crop can fail (len < max_seq + 2, see Data.__init__) on every rank alike
    dataset = Data(options.data_path, options.max_seq, field=options.field, rng=random.Random(1234 + rank),
                   min_length=options.max_seq + 2 if multi_gpu else None)
    log(dataset)
    dataset.check_vocab(vocab)        # out-of-vocabulary ids raise here (the kernels do not range-check per step)

    log('Loading model')
    mt = MusicTransformer(**model_config)
    mt.to(device)
    from .dp import DataParallel
    dp = DataParallel(mt, groups=options.buckets or None) if multi_gpu else None
    opt = FusedAdam(mt, lr=0, betas=(0.9, 0.98), eps=1e-9, grad_scale=dp.grad_scale if dp else 1.0)
    scheduler = CustomSchedule(config.embedding_dim, optimizer=opt)   # d_model of the schedule = config constant, as in train.py:144
    start_epoch = 0

    def to_dev(a):
        return torch.from_numpy(a).contiguous().to(device, non_blocking=True, dtype=torch.int)

    def evaluate():
        mt.eval()
        with torch.no_grad():
            ex, ey = dataset.slide_seq2seq_batch(2, options.max_seq, 'valid')
            try:                                           # the same host-side guard as on the training batches: early, and it names the phase
                utils.check_no_leading_pads(ex, pad)
            except ValueError as err:
                raise ValueError(f'evaluation batch: {err}') from err
            pred, _ = mt.forward(to_dev(ex))
            return metric_set(pred, to_dev(ey))

    if load_path is not None:
        checkpoint = torch.load(load_path, map_location=device, weights_only=False)
        mt.load_state_dict(checkpoint['net'])
        opt.load_state_dict(checkpoint['optimizer'])
        scheduler = CustomSchedule(config.embedding_dim, optimizer=opt)
        scheduler._step = int(checkpoint.get('sched_step', 0))     # extra key; the reference restarts warm-up
        start_epoch = checkpoint['epoch'] + 1
        log(f'Success load {load_path}')
        em = evaluate()
        log('Eval >>>> Loss: {:6.6}, Accuracy: {}'.format(em['loss'], em['accuracy']))
    log('-' * 70)

    def save_model(epoch, acc=0.0):
        if rank != 0:
            return
        os.makedirs(save_path, exist_ok=True)
        state = {'net': mt.state_dict(), 'optimizer': opt.state_dict(), 'epoch': epoch, 'sched_step': scheduler._step}
        log('Saving to', save_path + 'train-{}-{}.pth'.format(epoch, acc))
        torch.save(state, save_path + 'train-{}-{}.pth'.format(epoch, acc))
        log('Done saving')

    log(mt)
    opt.zero_grad()
    log(">> Train start...")
    idx = 0
    t_meter, ev_meter = time.time(), 0
    metrics = None
    e = start_epoch
    for e in range(start_epoch, options.epochs):
        try:
            log(">>> [Epoch was updated]")
            nb = max(1, len(dataset.file_dict['train']) // options.batch_size)
            b = 0
            for b in range(nb):
                try:
                    batch_x, batch_y = dataset.slide_seq2seq_batch(options.batch_size, options.max_seq)
                except (IndexError, ValueError):
                    # the reference swallows IndexError (train.py:261-262); a file of exactly max_seq+1 events raises
                    # ValueError from randrange instead.  Single process: skip like the reference.
                    if dp is not None and dp.world > 1:
                        raise RuntimeError("a rank failed to draw a batch under data parallelism: the ranks would stop "
                                           "issuing the same collectives (dataset changed after the min_length filter?)")
                    continue
                # host-side, before the H2D copy, and OUTSIDE the try above: its ValueError (which rows, why) must reach the user
                # instead of being taken for a failed draw (skipped silently / reported as a rank mismatch under DP)
                utils.check_no_leading_pads(batch_x, pad)
                batch_x, batch_y = to_dev(batch_x), to_dev(batch_y)
                mt.train()
                last_micro = (b + 1) % options.accum_grad == 0
                ctx = dp.no_sync() if (dp is not None and not last_micro) else _null()
                with ctx:
                    sample = mt.forward(batch_x)
                    metrics = metric_set(sample, batch_y)
                    loss = metrics['loss'] / options.accum_grad
                    if dp is not None and dp.world > 1 and metric_set.last_nonpad is not None:
                        loss = loss * dp.loss_weight(metric_set.last_nonpad)     # exact global-batch mean on padded data
                    loss.backward()
                if last_micro:
                    scheduler.step()
                    opt.zero_grad()
                idx += 1
                ev_meter += batch_x.numel() * world
                if options.max_batches and idx >= options.max_batches:
                    break
            eval_metrics = evaluate()
            if (e + 1) % options.saving_interval == 0:
                save_model(e, eval_metrics['accuracy'])
            # the device-side record of the leading-pads guard (this is a synchronisation point anyway).  AFTER the checkpoint: the
            # flag is sticky and shared by training and evaluation forwards, so raising first would lose the epoch just trained
            mt.check_no_leading_pads()
            dt = time.time() - t_meter
            log('\n====================================================')
            log('Epoch/Batch: {}/{}'.format(e, b))
            if metrics is not None:
                log('Train >>>> Loss: {:6.6}, Accuracy: {}'.format(metrics['loss'], metrics['accuracy']))
            log('Eval >>>> Loss: {:6.6}, Accuracy: {}'.format(eval_metrics['loss'], eval_metrics['accuracy']))
            log('Throughput >>>> {:.0f} events/s'.format(ev_meter / max(dt, 1e-9)))
            t_meter, ev_meter = time.time(), 0
            if options.max_batches and idx >= options.max_batches:
                break
        except KeyboardInterrupt:
            save_model(e)
            log(Exception)
            break
    save_model(options.epochs if not options.max_batches else e)
    if world > 1:
        dist.destroy_process_group()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


if __name__ == '__main__':
    main()
