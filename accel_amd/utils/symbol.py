"""Base of the network-description classes.

Contract kept from the reference's lib/utils/symbol.py:9-55 because harness
code relies on it: a `sym` attribute (also reachable as `.symbol`), and after
`infer_shape(shapes)` the three dicts `arg_shape_dict`, `out_shape_dict`,
`aux_shape_dict`; `check_parameter_shapes` asserts that every parameter of the
graph is present with the inferred shape ("<name> not initialized" /
"shape inconsistent for <name> ...")."""
import math


class Symbol(object):
    sym = None
    arg_shape_dict = out_shape_dict = aux_shape_dict = None

    symbol = property(lambda self: self.sym)

    # -- to be provided by the model classes ------------------------------------------------
    def get_symbol(self, cfg, is_train=True):
        raise NotImplementedError("%s does not build a graph" % type(self).__name__)

    def init_weights(self, cfg, arg_params, aux_params):
        raise NotImplementedError("%s has no initialiser" % type(self).__name__)

    # -- helpers ---------------------------------------------------------------------------------
    @staticmethod
    def get_msra_std(shape):
        """He/MSRA standard deviation sqrt(2 / fan_in), fan_in = Cin * kh * kw."""
        fan_in = 1.0
        for d in shape[1:]:
            fan_in *= d
        return math.sqrt(2.0 / fan_in)

    def infer_shape(self, data_shape_dict):
        args, outs, auxs = self.sym.infer_shape(**data_shape_dict)
        names = (self.sym.list_arguments(), self.sym.list_outputs(), self.sym.list_auxiliary_states())
        self.arg_shape_dict, self.out_shape_dict, self.aux_shape_dict = (
            dict(zip(n, s)) for n, s in zip(names, (args, outs, auxs)))

    def check_parameter_shapes(self, arg_params, aux_params, data_shape_dict, is_train=True):
        def verify(name, given, expected):
            assert name in given, name + ' not initialized'
            got = tuple(given[name].shape)
            assert got == tuple(expected[name]), \
                'shape inconsistent for %s inferred %s provided %s' % (name, expected[name], got)

        for name in self.sym.list_arguments():
            is_input = name in data_shape_dict
            is_test_label = (not is_train) and 'label' in name
            if not (is_input or is_test_label):
                verify(name, arg_params, self.arg_shape_dict)
        for name in self.sym.list_auxiliary_states():
            verify(name, aux_params, self.aux_shape_dict)
