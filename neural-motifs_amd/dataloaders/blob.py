"""
Per-step batch container with the reference's interface (dataloaders/blob.py): `append(entry)`, `reduce()`,
`scatter()`, `blob[gpu] -> (imgs, im_sizes, image_offset, gt_boxes, gt_classes, gt_rels, proposals[, train_anchor_inds])`.

Data parallelism is one process per GPU, so a Blob always describes ONE device's images (`num_gpus == 1`,
`image_offset == 0`); the ragged torch.nn.parallel scatter of the reference (:148-153) has no counterpart.
Anchor targets (consumed only by detector pre-training, models/train_detector.py) are produced for mode 'det' exactly
as the reference does while collating (blob.py:91-102): `train_anchors` [k,8] (anchor, matched GT box),
`train_anchor_labels` [k,5] (img, h, w, A, label), `train_anchor_inds` = its first four columns.  In 'rel' mode, where
nothing reads them, `train_anchor_inds` is an empty [0,4] tensor.  `anchor_rs` injects the sampler's RNG.
"""
import numpy as np
import torch

from lib.pytorch_misc import set_host

_COPY_STREAMS = {}          # device index -> the stream Blob.prefetch() copies on


class Blob(object):
    def __init__(self, mode='det', is_train=False, num_gpus=1, primary_gpu=None, batch_size_per_gpu=3):
        assert mode in ('det', 'rel')
        if num_gpus != 1:
            raise ValueError('one process per GPU: build one Blob(num_gpus=1) per rank')
        self.mode = mode
        self.is_train = is_train
        self.num_gpus = 1
        self.batch_size_per_gpu = batch_size_per_gpu
        self.primary_gpu = primary_gpu
        self.imgs, self.im_sizes = [], []
        self.gt_boxes, self.gt_classes, self.gt_rels = [], [], []
        self.proposals = []
        self.train_anchor_labels, self.train_anchors = [], []
        self.train_anchor_inds = None
        self.anchor_rs = None
        self.proposal_chunks = None

    @property
    def is_rel(self):
        return self.mode == 'rel'

    def append(self, d):
        """add one image entry (the dict of dataloaders/visual_genome.py:187-197)"""
        i = len(self.imgs)
        self.imgs.append(d['img'])
        h, w, scale = d['img_size']
        self.im_sizes.append((h, w, scale))
        self.gt_boxes.append(d['gt_boxes'].astype(np.float32) * d['scale'])
        self.gt_classes.append(np.column_stack((i * np.ones(d['gt_classes'].shape[0], dtype=np.int64), d['gt_classes'])))
        if self.is_rel:
            self.gt_rels.append(np.column_stack((i * np.ones(d['gt_relations'].shape[0], dtype=np.int64),
                                                 d['gt_relations'])))
        if self.is_train and not self.is_rel:
            from lib.fpn.anchor_targets import anchor_target_layer
            anchors_, inds_, targets_, labels_ = anchor_target_layer(self.gt_boxes[-1], (h, w), rs=self.anchor_rs)
            self.train_anchors.append(np.hstack((anchors_, targets_)))
            self.train_anchor_labels.append(np.column_stack((i * np.ones(inds_.shape[0], dtype=np.int64), inds_, labels_)))
        if 'proposals' in d:
            self.proposals.append(np.column_stack((i * np.ones(d['proposals'].shape[0], dtype=np.float32),
                                                   d['scale'] * d['proposals'].astype(np.float32))))

    def reduce(self):
        if len(self.imgs) != self.batch_size_per_gpu:
            raise ValueError("Wrong batch size? imgs len {} bsize/gpu {}".format(len(self.imgs), self.batch_size_per_gpu))
        self.imgs = torch.stack(self.imgs, 0)
        self.im_sizes = np.stack(self.im_sizes).reshape((1, self.batch_size_per_gpu, 3))
        if self.is_rel:
            self.gt_rels = torch.from_numpy(np.concatenate(self.gt_rels, 0)).long()
        self.gt_boxes = torch.from_numpy(np.concatenate(self.gt_boxes, 0)).float()
        self.gt_classes = torch.from_numpy(np.concatenate(self.gt_classes, 0)).long()
        if self.is_train and not self.is_rel:
            self.train_anchor_labels = torch.from_numpy(np.concatenate(self.train_anchor_labels, 0)).long()
            self.train_anchors = torch.from_numpy(np.concatenate(self.train_anchors, 0)).float()
            self.train_anchor_inds = self.train_anchor_labels[:, :-1].contiguous()
        elif self.is_train:
            self.train_anchor_inds = torch.zeros(0, 4, dtype=torch.long)
        if len(self.proposals) != 0:
            self.proposals = torch.from_numpy(np.concatenate(self.proposals, 0)).float()
            self.proposal_chunks = [self.proposals.shape[0]]

    def pin_memory(self):
        """page-lock the batch (torch's DataLoader calls this in its pinning thread when `pin_memory=True`): `scatter()` is
        then a set of asynchronous DMA copies.  From pageable memory every copy first waits for the GPU to drain its queue,
        which serialises the host's launch work of the step behind the previous step's kernels."""
        for name in ('imgs', 'gt_boxes', 'gt_classes', 'gt_rels', 'train_anchor_inds', 'train_anchor_labels',
                     'train_anchors', 'proposals'):
            t = getattr(self, name, None)
            if torch.is_tensor(t) and not t.is_cuda and t.numel() > 0:
                setattr(self, name, t.pin_memory())
        return self

    def _to_device(self, x, mirror=False):
        # one process per GPU: the target is the device THIS rank made current (torch.cuda.set_device(local_rank) in
        # lib/dist.init_from_env / the drivers), unless a caller pins primary_gpu explicitly (reference blob.py:20)
        dev = torch.cuda.current_device() if self.primary_gpu is None else self.primary_gpu
        y = x.cuda(dev, non_blocking=True)
        if mirror:                       # small GT index arrays keep their host values (lib/pytorch_misc.py: host mirrors)
            set_host(y, x.numpy())
        return y

    _TENSORS = ('imgs', 'gt_classes', 'gt_boxes', 'gt_rels', 'train_anchor_inds', 'train_anchor_labels', 'train_anchors', 'proposals')

    def prefetch(self):
        """start the batch's host -> HBM copies NOW, on this device's copy stream, without touching the compute streams (round 6).
        The loop hands the NEXT batch here while the current step's kernels run (models/train_rels.py, bench.py); `scatter()` then
        only makes the consuming stream wait for the copies' event.  A 25 MB batch is 0.5 ms of DMA: on the compute stream it sat
        in front of the trunk of every step (bench.py `h2d_inclusive`).  Page-locked batches only (Blob.pin_memory); no-op on a
        batch that is already on the device or already on its way."""
        if getattr(self, '_prefetch_event', None) is not None or not torch.cuda.is_available():
            return self
        if not isinstance(self.imgs, torch.Tensor) or self.imgs.is_cuda or not self.imgs.is_pinned():
            return self
        dev = torch.cuda.current_device() if self.primary_gpu is None else self.primary_gpu
        st = _COPY_STREAMS.get(dev)
        if st is None:
            st = _COPY_STREAMS[dev] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            self._scatter_now()
            ev = torch.cuda.Event()
            ev.record(st)
        self._prefetch_event = ev
        return self

    def scatter(self):
        """move the batch to this process's GPU (asynchronous H2D); after `prefetch()`: wait for the copies that are already in flight"""
        ev = getattr(self, '_prefetch_event', None)
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for name in self._TENSORS:
                t = getattr(self, name, None)
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)            # allocated under the copy stream, consumed here
            self._prefetch_event = None
            return
        if isinstance(self.imgs, torch.Tensor) and self.imgs.is_cuda:
            return
        self._scatter_now()

    def _scatter_now(self):
        self.imgs = self._to_device(self.imgs)
        self.gt_classes = self._to_device(self.gt_classes, mirror=True)
        self.gt_boxes = self._to_device(self.gt_boxes, mirror=True)
        if self.is_rel:
            self.gt_rels = self._to_device(self.gt_rels, mirror=True)
        if self.is_train:
            self.train_anchor_inds = self._to_device(self.train_anchor_inds)
            if not self.is_rel:
                self.train_anchor_labels = self._to_device(self.train_anchor_labels)
                self.train_anchors = self._to_device(self.train_anchors)
        if self.proposal_chunks is not None:
            self.proposals = self._to_device(self.proposals)

    def __getitem__(self, index):
        if index != 0:
            raise ValueError("Out of bounds with index {} and 1 gpu per process".format(index))
        rels = self.gt_rels if self.is_rel else None
        proposals = self.proposals if self.proposal_chunks is not None else None
        if self.is_train:
            return (self.imgs, self.im_sizes[0], 0, self.gt_boxes, self.gt_classes, rels, proposals,
                    self.train_anchor_inds)
        return self.imgs, self.im_sizes[0], 0, self.gt_boxes, self.gt_classes, rels, proposals
