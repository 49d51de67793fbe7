"""
`VG` / `VGDataLoader` entry points of the reference (dataloaders/visual_genome.py:23-424) for the drivers.

The Visual Genome HDF5/JSON files are not part of this environment (out of scope for the hot path),
so `VG.splits` serves the synthetic VG-shaped dataset (dataloaders/synthetic.py) -- same entry dict, same attributes
(`ind_to_classes`, `ind_to_predicates`, `gt_classes`, `gt_boxes`, `relationships`).  Real-data loading is listed as
the next row of the scope table.
"""
import os

from config import VG_SGG_FN
from dataloaders.synthetic import SyntheticVG, SyntheticLoader


class VG(SyntheticVG):
    @classmethod
    def splits(cls, num_val_im=5000, filter_duplicate_rels=True, use_proposals=False, filter_non_overlap=False,
               num_train_im=None, seed=1234, **kwargs):
        if os.path.exists(VG_SGG_FN):
            raise NotImplementedError('reading VG-SGG.h5 is not built yet (SURVEY.md §8f rank 2); '
                                      'unset MOTIFS_DATA to use the synthetic stand-in')
        n_train = num_train_im if num_train_im is not None else 96
        n_val = max(1, min(num_val_im, 24))
        return cls(num_images=n_train, seed=seed), cls(num_images=n_val, seed=seed + 1), cls(num_images=n_val, seed=seed + 2)


class VGDataLoader(SyntheticLoader):
    @classmethod
    def splits(cls, train_data, val_data, batch_size=3, num_workers=1, num_gpus=1, mode='det', rank=0, world_size=1,
               **kwargs):
        assert mode in ('det', 'rel')
        train = cls(train_data, batch_size, True, rank=rank, world_size=world_size, mode=mode)
        val = cls(val_data, batch_size if mode == 'det' else 1, False, rank=rank, world_size=world_size, mode=mode)
        return train, val
