"""Boxes / Instances / ImageList with the subset of the Detectron2 API the hot path uses."""
import itertools

import torch


class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def to(self, *a, **k):
        return Boxes(self.tensor.to(*a, **k))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def clone(self):
        return Boxes(self.tensor.clone())

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"


class Instances:
    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        ret = Instances(self._image_size)
        for key, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*a, **k)
            ret.set(key, v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists):
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __repr__(self):
        s = self.__class__.__name__ + "("
        s += "num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self) if self._fields else 0, self._image_size[0], self._image_size[1],
            ", ".join("{}: {}".format(k, v) for k, v in self._fields.items()))
        return s


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        """NCHW padded batch (host/oracle use; the product path pads inside utv2_preprocess_image)."""
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        hm, wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            hm, wm = (hm + d - 1) // d * d, (wm + d - 1) // d * d
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], hm, wm), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out, sizes)
