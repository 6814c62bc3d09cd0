"""Run-wide values that models read without a handle on the engine — reference passl/utils/infohub.py:17-30
(``runtime_info_hub.max_steps`` feeds CosineEMA's momentum schedule, passl/models/utils/averaged_model.py:178-180;
the Engine fills it in ``init_runtime_info_hub``, passl/engine/engine.py:346-349)."""


class RuntimeInfoHub(dict):
    def __getattr__(self, key):
        if key not in self:
            raise ValueError('`{0}` not in RuntimeInfoHub, please set it firstly by `runtime_info_hub.{0} = value`, '
                             'e.g. `from passl.utils.infohub import runtime_info_hub; '
                             'runtime_info_hub.max_steps = 10000`'.format(key))
        return self[key]

    def __setattr__(self, key, value):
        self[key] = value


runtime_info_hub = RuntimeInfoHub()
