"""Drop-in `MyDataloader` module: the batch layout the train / validate loops consume.

PathDataset is the reference's list-of-path-ids dataset (src/MyDataloader.py:62-73) for
torch.utils.data.DataLoader(batch_size=1350, shuffle=True) (src/train.py:469-472); it holds python ints only,
so nothing here touches the GPU.  The unused cone sampler of the reference (src/MyDataloader.py:4-59, dead code
with debug prints) is not reproduced.
"""
from torch.utils.data import Dataset

__all__ = ['PathDataset']


class PathDataset(Dataset):
    def __init__(self, paths):
        self.paths = paths

    def __getitem__(self, index):
        return self.paths[index]

    def __len__(self):
        return len(self.paths)
