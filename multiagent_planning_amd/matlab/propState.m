function [p,v] = propState(po, a, A_p, A_v, K)
% Shadows dec-iSCP/propState.m (same signature): p = [po'; A_p a + repmat(po')], v = [0; A_v a], on the GPU.
prm = dmpc_params_struct(0, 0.2, 15, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);      % context only
[new_p,new_v] = dmpc_mex('prop_state', prm, A_p, A_v, [], po(:), [0;0;0], a(:));
p = [po(:); new_p];
v = [0;0;0; new_v];
end
