"""parl_amd.remote — the @parl.remote_class / parl.connect API surface
(parl/remote/remote_decorator.py:25-113, parl/remote/client.py) bound to IN-PROCESS objects.

The reference's xparl is a ZeroMQ/gRPC cluster that ships the decorated class to CPU job
processes; rebuilding that control plane is out of scope (§8).  What
the IMPALA/A2C scripts need from it is the calling convention, and that is what is kept:
  * `@remote_class` / `@remote_class(wait=False, max_memory=..., n_gpu=...)`;
  * env XPARL=True makes the decorator return the undecorated class (remote_decorator.py:75-77);
  * wait=True: method calls run synchronously and return their value;
  * wait=False: every call returns a FutureObject whose .get() can be consumed once
    (future_mode/future_object.py:57-85), executed by one daemon worker thread per instance
    (proxy_wrapper_nowait.py:78-81) in call order;
  * attribute get/set is proxied (proxy_wrapper.py:69-76); exceptions in remote calls surface
    as RemoteError.
A GPU-resident actor cannot live in a CPU job process anyway (job.py:17 hides the GPUs), so the
in-process proxy is also the fast path: `sample()` returns device tensors with no pickling."""
import os
import queue
import threading

__all__ = ['remote_class', 'connect', 'disconnect', 'RemoteError', 'FutureObject', 'FutureGetRepeatedlyError',
           'FutureFunctionError']

_connected = {'address': None}


class RemoteError(Exception):
    def __init__(self, func_name, error_info):
        super(RemoteError, self).__init__('[remote error] in %s: %s' % (func_name, error_info))
        self.error_info = '[remote error] in %s: %s' % (func_name, error_info)


class FutureGetRepeatedlyError(Exception):
    pass


class FutureFunctionError(Exception):
    pass


def connect(master_address=None, distributed_files=None, recursive_watch=False):
    """parl.connect: records the address; there is no cluster to join (in-process actors)."""
    _connected['address'] = master_address


def disconnect():
    _connected['address'] = None


class FutureObject(object):
    def __init__(self, name):
        self._q = queue.Queue(maxsize=1)
        self._name = name
        self._consumed = False

    def _set(self, ok, value):
        self._q.put((ok, value))

    def get(self, block=True, timeout=None):
        if self._consumed:
            raise FutureGetRepeatedlyError('the result of %s has already been fetched' % self._name)
        ok, value = self._q.get(block=block, timeout=timeout)
        self._consumed = True
        if not ok:
            raise RemoteError(self._name, value)
        return value


class _NoWaitProxy(object):
    _RESERVED = ('_xparl_obj', '_xparl_q', '_xparl_thread')

    def __init__(self, cls, args, kwargs):
        object.__setattr__(self, '_xparl_obj', cls(*args, **kwargs))
        object.__setattr__(self, '_xparl_q', queue.Queue())
        t = threading.Thread(target=self._xparl_run, daemon=True)
        object.__setattr__(self, '_xparl_thread', t)
        t.start()

    def _xparl_run(self):
        while True:
            fut, fn, a, k = self._xparl_q.get()
            try:
                fut._set(True, fn(*a, **k))
            except Exception as e:  # noqa: BLE001 - reported through the future
                fut._set(False, repr(e))

    def __getattr__(self, name):
        if name.startswith('_xparl'):
            raise FutureFunctionError('attributes starting with _xparl are reserved')
        attr = getattr(self._xparl_obj, name)
        if not callable(attr):
            return attr

        def call(*a, **k):
            fut = FutureObject(name)
            self._xparl_q.put((fut, attr, a, k))
            return fut

        return call

    def __setattr__(self, name, value):
        setattr(self._xparl_obj, name, value)


class _WaitProxy(object):
    def __init__(self, cls, args, kwargs):
        object.__setattr__(self, '_xparl_obj', cls(*args, **kwargs))
        object.__setattr__(self, '_xparl_lock', threading.Lock())

    def __getattr__(self, name):
        attr = getattr(self._xparl_obj, name)
        if not callable(attr):
            return attr

        def call(*a, **k):
            with self._xparl_lock:  # RemoteWrapper.internal_lock (remote_wrapper.py:72,186)
                try:
                    return attr(*a, **k)
                except Exception as e:  # noqa: BLE001
                    raise RemoteError(name, repr(e))

        return call

    def __setattr__(self, name, value):
        setattr(self._xparl_obj, name, value)


def remote_class(*args, **kwargs):
    """Decorator with the reference's keyword arguments (remote_decorator.py:92-99)."""
    wait = kwargs.get('wait', True)

    def decorator(cls):
        if os.environ.get('XPARL') == 'True':  # inside a job: raw class (remote_decorator.py:75-77)
            return cls

        class RemoteProxy(object):
            _original = cls

            def __new__(klass, *a, **k):
                return (_WaitProxy if wait else _NoWaitProxy)(cls, a, k)

        RemoteProxy.__name__ = cls.__name__
        return RemoteProxy

    if len(args) == 1 and callable(args[0]) and not kwargs:
        return decorator(args[0])
    return decorator
