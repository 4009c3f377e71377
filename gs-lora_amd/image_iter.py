"""The torchvision-free part of the reference's `image_iter.py`: `CustomSubset` (:124-137), the Subset the few-shot sampler returns
(`util/utils.py:496`) and the continual driver splits its data sets into (`train_own_forget_cl.py:547-560`). The MXNet record readers and
the ImageFolder wrappers of that file are data plumbing outside the hot path (out of scope)."""
import torch


class CustomSubset(torch.utils.data.Subset):
    """Subset that keeps `targets` / `classes` of the parent (reference image_iter.py:124-137)."""

    def __init__(self, dataset, indices):
        super().__init__(dataset, indices)
        self.targets = dataset.targets
        self.classes = dataset.classes

    def __getitem__(self, idx):
        return self.dataset[self.indices[idx]]

    def __len__(self):
        return len(self.indices)
