"""DAGLayer: string modules together as a list-of-nodes DAG (mirror of ddsp/dags.py:57-195).

Host-side plumbing only (dict routing, no arithmetic); modules are any callables, Processors are
recognised by duck typing on get_signal/get_controls exactly as the reference does (dags.py:44).
"""
import logging

from ddsp_amd import core

# duck typing (ddsp/dags.py:40-44)
is_loss = lambda v: hasattr(v, 'get_losses_dict')
is_processor = lambda v: hasattr(v, 'get_signal') and hasattr(v, 'get_controls')
is_module = lambda v: callable(v) and hasattr(v, 'name') and not isinstance(v, str)


def split_keras_kwargs(kwargs):
  """Strip keras specific kwargs (ddsp/dags.py:47-53)."""
  keras_kwargs = {}
  for key in ['training', 'mask', 'name']:
    if kwargs.get(key) is not None:
      keras_kwargs[key] = kwargs.pop(key)
  return keras_kwargs, kwargs


class DAGLayer:
  """String modules together (ddsp/dags.py:57-195).

  dag: list of nodes ['module', ['input_key', ...], ['output_key', ...]]; 'module' is an instance
  or the name of a kwarg module; input keys are nested keys ("inputs/f0_hz", "harmonic/signal")
  into the running outputs dict; the graph is read sequentially (topologically sorted).
  """

  def __init__(self, dag, **kwarg_modules):
    keras_kwargs, kwarg_modules = split_keras_kwargs(kwarg_modules)
    self.name = keras_kwargs.get('name', type(self).__name__.lower())
    self.built = False
    modules = {k: v for k, v in kwarg_modules.items() if is_module(v)}
    dag, dag_modules = self.format_dag(dag)
    self.dag = dag
    modules.update(dag_modules)
    self.module_names = list(modules.keys())
    for module_name, module in modules.items():
      setattr(self, module_name, module)

  @property
  def modules(self):
    return [getattr(self, name) for name in self.module_names]

  @staticmethod
  def format_dag(dag):
    """Remove modules from dag, and replace with module names (ddsp/dags.py:112-127)."""
    modules = {}
    dag = list(dag)
    for i, node in enumerate(dag):
      node = list(node)
      module = node[0]
      if is_module(module):
        modules[module.name] = module
        node[0] = module.name
      dag[i] = node
    return dag, modules

  def __call__(self, inputs, **kwargs):
    return self.call(inputs, **kwargs)

  def call(self, inputs, **kwargs):
    return self.run_dag(inputs, **kwargs)

  def run_dag(self, inputs, verbose=False, **kwargs):
    """Connects and runs submodules of dag; returns the nested dict of all outputs (dags.py:134-195)."""
    outputs = {'inputs': inputs}
    outputs.update(inputs)          # reference keeps the inputs in the base namespace too
    module_outputs = None
    for node in self.dag:
      module_key, input_keys = node[0], node[1]
      module = getattr(self, module_key)
      output_keys = node[2] if len(node) > 2 else None
      node_inputs = [core.nested_lookup(key, outputs) for key in input_keys]
      if verbose:
        logging.info('Input to Module: %s\nKeys: %s\n', module_key, input_keys)
      if is_processor(module):
        module_outputs = module(*node_inputs, return_outputs_dict=True, **kwargs)
      elif is_loss(module):
        module_outputs = module.get_losses_dict(*node_inputs, **kwargs)
      else:
        module_outputs = module(*node_inputs, **kwargs)
      if not isinstance(module_outputs, dict):
        module_outputs = core.to_dict(module_outputs, output_keys)
      outputs[module_key] = module_outputs
    outputs['out'] = module_outputs   # 'out' is a reserved key for the final dag output
    return outputs
