"""Typed hyper-parameter container behind the JSON config schema.

Behavioural mirror of the reference's ``athena/utils/hparam.py`` (``HParams`` :305-686,
``register_and_parse_hparams`` :688-695) for the surface the training path uses: typed
name/value pairs registered from a ``default_config`` dict, overridden from a (JSON) dict with
``KeyError`` for unknown names and ``ValueError`` for values whose type does not fit, nested
dict values replaced wholesale.  Written from the behaviour, not from the reference's code.
"""
import json
import numbers
import re

_CLAUSE = re.compile(r"\s*(?P<name>[a-zA-Z][\w\.]*)\s*(\[\s*(?P<index>\d+)\s*\])?\s*=\s*"
                     r"(\[(?P<list>[^\]]*)\]|(?P<scalar>[^,\[]*))\s*(,|$)")


def _coerce(name, kind, value):
    """Cast ``value`` to the registered type ``kind`` if that loses nothing, else ValueError."""
    if kind is type(None):
        return value
    bad = ValueError("Could not cast hparam '%s' of type '%s' from value %r" % (name, kind, value))
    if issubclass(kind, (str, bytes)):
        if not isinstance(value, (str, bytes)):
            raise bad
    if issubclass(kind, bool) != isinstance(value, bool):
        raise bad
    if issubclass(kind, numbers.Integral) and not isinstance(value, numbers.Integral):
        raise bad
    if issubclass(kind, numbers.Number) and not isinstance(value, numbers.Number):
        raise bad
    return kind(value)


def _parse_scalar(name, kind, text, whole):
    text = text.strip()
    try:
        if kind is bool:
            if text in ("true", "True"):
                return True
            if text in ("false", "False"):
                return False
            return bool(int(text))
        return kind(text)
    except ValueError:
        raise ValueError("Could not parse hparam '%s' of type '%s' with value '%s' in %s" %
                         (name, kind.__name__, text, whole))


def parse_values(values, type_map, ignore_unknown=False):
    """Parse ``"a=1,b=[1,2],c[1]=x"`` into a dict using the registered element types."""
    out = {}
    pos = 0
    while pos < len(values):
        m = _CLAUSE.match(values, pos)
        if not m:
            raise ValueError("Malformed hyperparameter value: %s" % values[pos:])
        pos = m.end()
        name = m.group("name")
        if name not in type_map:
            if ignore_unknown:
                continue
            raise ValueError("Unknown hyperparameter type for %s" % name)
        kind = type_map[name]
        index = m.group("index")
        if m.group("list") is not None:
            if index is not None:
                raise ValueError("Assignment of a list to a list index.")
            if name in out:
                raise ValueError("Multiple assignments to variable '%s' in %s" % (name, values))
            out[name] = [_parse_scalar(name, kind, e, values) for e in re.split("[ ,]", m.group("list")) if e]
        elif m.group("scalar") is not None and m.group("scalar").strip() != "":
            val = _parse_scalar(name, kind, m.group("scalar"), values)
            if index is None:
                if name in out:
                    raise ValueError("Multiple assignments to variable '%s' in %s" % (name, values))
                out[name] = val
            else:
                slot = out.setdefault(name, {})
                if not isinstance(slot, dict) or int(index) in slot:
                    raise ValueError("Multiple assignments to variable '%s' in %s" % (name, values))
                slot[int(index)] = val
        else:
            raise ValueError("Could not parse hparam '%s' of type '%s' with value '' in %s" %
                             (name, kind.__name__, values))
    return out


class HParams(object):
    """Named, typed hyper-parameters accessible as attributes."""

    def __init__(self, model_structure=None, **kwargs):
        self._hparam_types = {}
        self._model_structure = model_structure
        for name, value in kwargs.items():
            self.add_hparam(name, value)

    def add_hparam(self, name, value):
        if getattr(self, name, None) is not None:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError("Multi-valued hyperparameters cannot be empty: %s" % name)
            self._hparam_types[name] = (type(value[0]), True)
        else:
            self._hparam_types[name] = (type(value), False)
        setattr(self, name, value)

    def set_hparam(self, name, value):
        kind, is_list = self._hparam_types[name]      # KeyError for an unknown name
        if isinstance(value, list):
            if not is_list:
                raise ValueError("Must not pass a list for single-valued parameter: %s" % name)
            setattr(self, name, [_coerce(name, kind, v) for v in value])
        else:
            if is_list:
                raise ValueError("Must pass a list for multi-valued parameter: %s." % name)
            setattr(self, name, _coerce(name, kind, value))

    def del_hparam(self, name):
        if hasattr(self, name):
            delattr(self, name)
            del self._hparam_types[name]

    def parse(self, values, ignore_unknown=False):
        type_map = {n: t for n, (t, _) in self._hparam_types.items()}
        if type(values) is dict:
            values = ",".join("%s=%s" % (k, v) for k, v in values.items())
        return self.override_from_dict(parse_values(values, type_map, ignore_unknown))

    def override_from_dict(self, values_dict):
        for name, value in values_dict.items():
            self.set_hparam(name, value)
        return self

    def set_model_structure(self, model_structure):
        self._model_structure = model_structure

    def get_model_structure(self):
        return self._model_structure

    def to_json(self, indent=None, separators=None, sort_keys=False):
        def strip(x):
            if isinstance(x, dict):
                return {k: strip(v) for k, v in x.items() if not callable(v)}
            if isinstance(x, list):
                return [strip(i) for i in x if not callable(i)]
            return x
        return json.dumps(strip(self.values()), indent=indent, separators=separators, sort_keys=sort_keys)

    def parse_json(self, values_json):
        return self.override_from_dict(json.loads(values_json))

    def values(self):
        return {n: getattr(self, n) for n in self._hparam_types}

    def get(self, key, default=None):
        if key in self._hparam_types:
            if default is not None:
                kind, is_list = self._hparam_types[key]
                msg = "Hparam '%s' of type '%s' is incompatible with default=%s" % (
                    key, ("list<%s>" % kind) if is_list else str(kind), default)
                if is_list != isinstance(default, list):
                    raise ValueError(msg)
                try:
                    for v in (default if is_list else [default]):
                        _coerce(key, kind, v)
                except ValueError as e:
                    raise ValueError("%s. %s" % (msg, e))
            return getattr(self, key)
        return default

    def __contains__(self, key):
        return key in self._hparam_types

    def __str__(self):
        return str(sorted(self.values().items()))

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, self.__str__())

    def instantiate(self):
        assert "cls" in self._hparam_types and self.cls is not None
        return self.cls(self)

    def append(self, hp):
        assert type(hp) == HParams
        for key, val in hp.values().items():
            if key != "cls":
                self.add_hparam(key, val)


def register_and_parse_hparams(default_config, config=None, **kwargs):
    """Register ``default_config`` then override from ``config`` (hparam.py:688-695)."""
    hparams = HParams(**kwargs)
    for key in default_config:
        hparams.add_hparam(key, default_config[key])
    if config is not None:
        hparams.override_from_dict(config)
    return hparams
