function Aaug = getPosVelMat(h,K)
% Shadows dec-iSCP/getPosVelMat.m (same signature): [final position; final velocity; last and first acceleration selectors] (12 x 3K).
prm = dmpc_params_struct(0, h, K, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);
Aaug = dmpc_mex('posvel_matrix', prm);
end
