"""Mirror of tiger/utils.py: prefetching wrappers around a batch iterator.

Only the thread variant is provided: collation uses the GPU, so it cannot live in a forked
worker process (the reference's BackgroundProcessGenerator is unused by its scripts as well)."""
import queue
import threading

_END = object()


class BackgroundThreadGenerator:
    """Iterates `generator` in a daemon thread, keeping up to `max_prefetch` items ready
    (tiger/utils.py:33-57).  An exception raised by the producer is re-raised in the consumer."""

    def __init__(self, generator, max_prefetch: int = 1):
        self._q = queue.Queue(max_prefetch)
        self._done = False
        self._thread = threading.Thread(target=self._produce, args=(generator,), daemon=True)
        self._thread.start()

    def _produce(self, generator):
        try:
            for item in generator:
                self._q.put(item)
            self._q.put(_END)
        except BaseException as e:  # surfaced by __next__
            self._q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration
        item = self._q.get()
        if item is _END:
            self._done = True
            raise StopIteration
        if isinstance(item, BaseException):
            self._done = True
            raise item
        return item
