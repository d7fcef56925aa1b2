"""Aspect-ratio grouped 4-tuple batcher of the semi-supervised loader (reference ubteacher/data/common.py:93-167).

Each element of the two input streams is a (strong dict, weak dict) pair of one image.  Landscape (w > h) and portrait images are
collected in separate buckets per stream; a batch is emitted when BOTH the labeled bucket in use holds `batch_size_label` pairs and the
unlabeled one `batch_size_unlabel`.  Reference behaviour kept on purpose: while one side waits for the other, the waiting side's
incoming elements are DROPPED (its full bucket is not touched, common.py:141-156), and the bucket a side "uses" is the one its most
recent accepted element went to."""


class _Side:
    def __init__(self, batch_size):
        self.batch_size = batch_size
        self.strong = ([], [])
        self.weak = ([], [])
        self.cur = None  # bucket id the side is currently filling / has filled

    def full(self):
        return self.cur is not None and len(self.strong[self.cur]) == self.batch_size

    def offer(self, pair):
        if self.full():
            return  # dropped, like the reference
        first = pair[0]
        self.cur = 0 if first["width"] > first["height"] else 1
        self.strong[self.cur].append(pair[0])
        self.weak[self.cur].append(pair[1])

    def take(self):
        s, w = self.strong[self.cur][:], self.weak[self.cur][:]
        del self.strong[self.cur][:]
        del self.weak[self.cur][:]
        return s, w


class AspectRatioGroupedSemiSupDatasetTwoCrop:
    """dataset = (labeled stream, unlabeled stream); batch_size = (labeled, unlabeled) per rank.  Iterating yields
    (label_strong, label_weak, unlabel_strong, unlabel_weak) lists of dicts (common.py:158-163)."""

    def __init__(self, dataset, batch_size):
        self.label_dataset, self.unlabel_dataset = dataset
        self.batch_size_label, self.batch_size_unlabel = batch_size[0], batch_size[1]
        self._label = _Side(self.batch_size_label)
        self._unlabel = _Side(self.batch_size_unlabel)

    def __iter__(self):
        for d_label, d_unlabel in zip(self.label_dataset, self.unlabel_dataset):
            self._label.offer(d_label)
            self._unlabel.offer(d_unlabel)
            if self._label.full() and self._unlabel.full():
                ls, lw = self._label.take()
                us, uw = self._unlabel.take()
                yield ls, lw, us, uw


class MapDataset:
    """Lazily maps an index stream over a list of dataset dicts (Detectron2 DatasetFromList + MapDataset + a sampler)."""

    def __init__(self, dicts, mapper, sampler):
        self.dicts, self.mapper, self.sampler = dicts, mapper, sampler

    def __len__(self):
        return len(self.dicts)

    def __iter__(self):
        for idx in self.sampler:
            out = self.mapper(self.dicts[idx])
            if out is not None:
                yield out
